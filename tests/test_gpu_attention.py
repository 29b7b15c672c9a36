"""GPU parity of the soft-attention decoder (st_attn_forward / st_attn_backward / st_attn_greedy)
against vectors produced by the reference's own RNN_Attn classes (tests/golden/attn_*_small.npz)."""
import numpy as np
import pytest
import torch
import torch.nn as nn

from oracle import restatement as R
from tests._util import load_fixture
from tests.test_gpu_decoder import _rel

pytestmark = pytest.mark.gpu
VOCAB = lambda w: {"<pad>": 0, "<start>": 1, "<end>": 2, "<unk>": 3}[w]   # vocab_builder.py:66-69


def _make(cell, params, dtype):
    from showtell_amd.rnn_attn import RNN_Attn
    from showtell_amd.rnn_attn_LSTM import RNN_Attn as RNN_Attn_LSTM
    V, E = params["embeddings.weight"].shape
    H = params["unit.weight_hh_l0"].shape[1]
    A, Fd = params["attn.encoder_att.weight"].shape
    L = R.num_layers_of(params)
    m = (RNN_Attn if cell == "gru" else RNN_Attn_LSTM)(E, Fd, A, H, V, L, dtype=dtype)
    m.load_state_dict(params)
    return m.cuda()


@pytest.mark.parametrize("route", ["dropin", "fused"])
@pytest.mark.parametrize("cell,name", [("gru", "attn_gru_small.npz"), ("lstm", "attn_lstm_small.npz")])
def test_fp32_attention_forward_loss_grads_match_reference_golden(cell, name, route):
    params, grads, d = load_fixture(name)
    m = _make(cell, params, torch.float32)
    feat = torch.from_numpy(d["feat"]).cuda()
    cap, lens = torch.from_numpy(d["caption"]).cuda(), d["lens"].tolist()
    alpha_c = float(d["alpha_c"])
    if route == "dropin":      # exactly Attention/main_attn.py:126-133
        target = nn.utils.rnn.pack_padded_sequence(cap, lens, batch_first=True)[0]
        logits, alphas = m(feat, cap, lens)
        assert _rel(logits, torch.from_numpy(d["logits"])) < 2e-4
        assert alphas.shape == tuple(d["alphas"].shape) and _rel(alphas, torch.from_numpy(d["alphas"])) < 2e-4
        loss = nn.CrossEntropyLoss()(logits, target)
        loss = loss + alpha_c * ((1. - alphas.sum(dim=1)) ** 2).mean()
    else:
        loss = m.loss(feat, cap, lens, alpha_c)
    assert abs(loss.item() - float(d["loss"])) < 3e-5
    loss.backward()
    named = dict(m.named_parameters())
    for k, g in grads.items():
        if k == "attn.full_att.bias":
            # softmax is shift invariant: d(loss)/d(full_att.bias) is analytically ZERO (the reference holds fp32 noise)
            assert named[k].grad.abs().max().item() < 1e-5 and g.abs().max().item() < 1e-5
            continue
        assert _rel(named[k].grad, g) < 3e-4, k


@pytest.mark.parametrize("cell,name", [("gru", "attn_gru_small.npz"), ("lstm", "attn_lstm_small.npz")])
def test_fp32_attention_greedy_ids_exact(cell, name):
    params, _, d = load_fixture(name)
    m = _make(cell, params, torch.float32).eval()
    ids = m.sentence_index(torch.from_numpy(d["feat"]).cuda(), VOCAB)
    assert ids.shape == tuple(d["greedy"].shape)
    assert np.array_equal(ids.cpu().numpy(), d["greedy"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_config3_shape_vs_oracle(dtype):
    """BASELINE config 3 geometry (F=2048, P=49, A=512, E=H=512) at a reduced batch / vocabulary / depth."""
    E, Fd, A, H, V, L, B = 512, 2048, 512, 512, 1000, 2, 8
    params = R.init_decoder_params(E, H, V, L, "gru", seed=9, attn=dict(F=Fd, A=A))
    if dtype == torch.bfloat16:
        params = {k: v.bfloat16().float() for k, v in params.items()}
    m = _make("gru", params, dtype)
    cap, lens = R.synthetic_captions(B, V, seed=9, mean=8, std=2, lo=4, hi=12)
    feat = torch.randn(B, Fd, 49, generator=torch.Generator().manual_seed(9)).abs()     # post-ReLU features are >= 0
    if dtype == torch.bfloat16:
        feat = feat.bfloat16().float()
    po = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    lo, logits_o, alphas_o = R.attn_train_loss(po, feat, cap, lens, 1.0, "gru")
    lo.backward()
    tol = 5e-4 if dtype == torch.float32 else 5e-2
    logits, alphas = m(feat.cuda(), cap.cuda(), lens)
    assert _rel(logits, logits_o) < tol and _rel(alphas, alphas_o) < tol
    # every alpha row of a live step sums to one, padded steps stay zero (rnn_attn.py:65,73)
    s = alphas.sum(2).cpu()
    for b, l in enumerate(lens):
        assert torch.allclose(s[b, :l], torch.ones(l), atol=1e-3) and float(s[b, l:].abs().sum()) == 0.0
    for p_ in m.parameters():
        p_.grad = None
    loss = m.loss(feat.cuda(), cap.cuda(), lens, 1.0)
    assert abs(loss.item() - lo.item()) < (1e-4 if dtype == torch.float32 else 3e-2)
    loss.backward()
    for k, p_ in m.named_parameters():
        if k == "attn.full_att.bias":
            assert p_.grad.abs().max().item() < (1e-5 if dtype == torch.float32 else 1e-2)
            continue
        assert _rel(p_.grad, po[k].grad) < (1e-3 if dtype == torch.float32 else 6e-2), k


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_full_config3_size(dtype):
    """BASELINE configs[2] at its FULL size (Attention/main_attn.py:123-134, rnn_attn.py:60-118): B = 64, L = 5, V = 10000,
    F = 2048, P = 49, A = E = H = 512, alpha_c = 1.  The whole batch is too slow for the CPU oracle inside the GPU suite, so:
    the 8 longest captions' logits rows and alphas against the oracle run on those 8 samples alone (rows are independent
    in the forward), plus the properties of the full batch: alphas rows sum to one on live steps and are zero on padded
    ones, loss = CE + alpha_c * mean((1 - sum_t alpha)^2) recomputed from the returned logits / alphas, finite gradients
    on every parameter, and a second SGD step on the same batch lowers the loss."""
    from showtell_amd import optim
    E, Fd, A, H, V, L, B = 512, 2048, 512, 512, 10000, 5, 64
    params = R.init_decoder_params(E, H, V, L, "gru", seed=21, attn=dict(F=Fd, A=A))
    if dtype == torch.bfloat16:
        params = {k: v.bfloat16().float() for k, v in params.items()}
    m = _make("gru", params, dtype).train()
    cap, lens = R.synthetic_captions(B, V, seed=21)
    feat = torch.randn(B, Fd, 49, generator=torch.Generator().manual_seed(21)).abs()
    if dtype == torch.bfloat16:
        feat = feat.bfloat16().float()
    logits, alphas = m(feat.cuda(), cap.cuda(), lens)
    ntok = sum(lens)
    assert logits.shape == (ntok, V) and alphas.shape == (B, lens[0], 49)
    s = alphas.sum(2).cpu()
    for b, l in enumerate(lens):
        assert torch.allclose(s[b, :l], torch.ones(l), atol=2e-3) and float(s[b, l:].abs().sum()) == 0.0
    # the 8 longest captions against the oracle
    S = 8
    with torch.no_grad():
        _, lo8, al8 = R.attn_train_loss({k: v.clone() for k, v in params.items()}, feat[:S], cap[:S, :lens[0]], lens[:S], 1.0, "gru")
    bs_full = [sum(1 for l in lens if l > t) for t in range(lens[0])]
    bs8 = [sum(1 for l in lens[:S] if l > t) for t in range(lens[0])]
    rows_full, rows8, of, o8 = [], [], 0, 0
    for t in range(lens[0]):
        for b in range(bs8[t]):
            rows_full.append(of + b); rows8.append(o8 + b)
        of += bs_full[t]; o8 += bs8[t]
    tol = 5e-4 if dtype == torch.float32 else 6e-2
    got = logits[torch.tensor(rows_full, device="cuda")].float().cpu()
    assert _rel(got, lo8[torch.tensor(rows8)]) < tol
    assert _rel(alphas[:S].float().cpu(), al8) < tol
    # loss of the fused route = CE(logits) + alpha_c * mean((1 - sum_t alpha)^2) of what forward() returned (main_attn.py:128-131)
    target = nn.utils.rnn.pack_padded_sequence(cap.cuda(), lens, batch_first=True)[0]
    want = nn.functional.cross_entropy(logits.float(), target) + ((1.0 - alphas.float().sum(1)) ** 2).mean()
    opt = optim.SGD(list(m.parameters()), lr=0.05, momentum=0.9, shadow_dtype=None if dtype == torch.float32 else dtype)
    losses = []
    for _ in range(3):
        opt.zero_grad()
        loss = m.loss(feat.cuda(), cap.cuda(), lens, 1.0)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert abs(losses[0] - float(want)) < (2e-4 if dtype == torch.float32 else 3e-2) * max(1.0, float(want))
    assert abs(losses[0] - np.log(V)) < 1.0 and losses[-1] < losses[0], losses
    for k, p_ in m.named_parameters():
        assert p_.grad is not None and torch.isfinite(p_.grad).all(), k
