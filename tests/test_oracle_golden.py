"""The oracle (oracle/restatement.py) against the committed golden vectors.

The vectors were produced by running the reference's own classes
(oracle/gen_golden.py); this pins the CPU restatement everywhere, including on
the GPU box where /root/reference does not exist.
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import restatement as R
from tests._util import GOLDEN, load_fixture, req_grad

ATOL = 2e-5  # fp32, different summation order than torch's fused RNN kernels


@pytest.mark.parametrize("cell,name", [("gru", "gru_small.npz"), ("lstm", "lstm_small.npz")])
def test_rnn_forward_loss_grads(cell, name):
    params, grads, d = load_fixture(name)
    p = req_grad(params)
    feat, cap, lens = torch.from_numpy(d["feat"]), torch.from_numpy(d["caption"]), d["lens"].tolist()
    loss, logits, target = R.gru_train_loss(p, feat, cap, lens, cell)
    assert torch.equal(target, torch.from_numpy(d["target"]))
    np.testing.assert_allclose(logits.detach().numpy(), d["logits"], atol=ATOL)
    assert abs(loss.item() - float(d["loss"])) < 1e-5
    loss.backward()
    for k, g in grads.items():
        np.testing.assert_allclose(p[k].grad.numpy(), g.numpy(), atol=ATOL, err_msg=k)


@pytest.mark.parametrize("cell,name", [("gru", "gru_small.npz"), ("lstm", "lstm_small.npz")])
def test_rnn_greedy_ids_exact(cell, name):
    params, _, d = load_fixture(name)
    feat = torch.from_numpy(d["feat"])
    with torch.no_grad():
        ids = R.rnn_greedy(params, feat, cell)
        ids1 = R.rnn_greedy(params, feat[:1], cell)
    assert ids.shape == (feat.shape[0], 25) and ids1.shape == (25,)  # rnn.py:56 squeeze
    assert np.array_equal(ids.numpy(), d["greedy"])
    assert np.array_equal(ids1.numpy(), d["greedy_b1"])


@pytest.mark.parametrize("k", [1, 3, 5])
def test_rnn_quirky_beam_ids_exact(k):
    params, _, d = load_fixture("gru_small.npz")
    feat = torch.from_numpy(d["feat"])[:1]
    with torch.no_grad():
        ids = R.rnn_beam_quirky(params, feat, k)
    assert np.array_equal(ids.numpy(), d[f"qbeam{k}"])
    if k == 1:  # rnn.py:43 claim: beam_size=1 == greedy
        assert np.array_equal(ids.numpy(), d["greedy_b1"])


@pytest.mark.parametrize("cell,name", [("gru", "attn_gru_small.npz"), ("lstm", "attn_lstm_small.npz")])
def test_attn_forward_loss_grads_greedy(cell, name):
    params, grads, d = load_fixture(name)
    p = req_grad(params)
    feat, cap, lens = torch.from_numpy(d["feat"]), torch.from_numpy(d["caption"]), d["lens"].tolist()
    loss, logits, alphas = R.attn_train_loss(p, feat, cap, lens, float(d["alpha_c"]), cell)
    np.testing.assert_allclose(logits.detach().numpy(), d["logits"], atol=ATOL)
    np.testing.assert_allclose(alphas.detach().numpy(), d["alphas"], atol=ATOL)
    assert abs(loss.item() - float(d["loss"])) < 1e-5
    loss.backward()
    for k, g in grads.items():
        np.testing.assert_allclose(p[k].grad.numpy(), g.numpy(), atol=ATOL, err_msg=k)
    with torch.no_grad():
        ids = R.attn_greedy(params, feat, 1, cell)
    assert np.array_equal(ids.numpy(), d["greedy"])


@pytest.mark.parametrize("bw,nh", [(5, 3), (4, 1)])
def test_beam_search_sequences_exact(bw, nh):
    params, _, d = load_fixture("beam_small.npz")
    feat = torch.from_numpy(d["feat"])
    ml = int(d[f"bw{bw}_maxlen"])
    for b in range(feat.shape[0]):
        with torch.no_grad():
            init, gen = R.gru_beam_callbacks(params, feat[b])
            hyp = R.beam_search(init, gen, [0], 1, 2, beam_width=bw, num_hypotheses=nh, max_length=ml)
        lens = d[f"bw{bw}_len"][b]
        assert len(hyp) == int((lens > 0).sum())  # may be [] (beam_search.py:69-79)
        for i, h in enumerate(hyp):
            v = h.to_sequence_of_values()
            assert v == d[f"bw{bw}_seq"][b, i, :lens[i]].tolist()
            assert abs(h.cum_cost - d[f"bw{bw}_cost"][b, i]) < 1e-4


def test_bleu_matches_reference_scorer():
    with open(os.path.join(GOLDEN, "bleu_small.json")) as f:
        j = json.load(f)
    got = R.bleu_corpus(j["gts"], j["res"], 4)
    np.testing.assert_allclose(got, j["bleu"], rtol=1e-12)


def test_pack_rows_and_batch_sizes_edge_cases():
    assert R.batch_sizes([3, 3, 1]) == [3, 2, 2]
    x = torch.arange(6).view(3, 2)
    assert R.pack_rows(x, [2, 1, 1]).tolist() == [0, 2, 4, 1]
    with pytest.raises(AssertionError):
        R.batch_sizes([1, 2])


def test_create_batch_contract():
    data = [("a", torch.zeros(3, 4, 4), torch.tensor([1, 5, 2])),
            ("b", torch.ones(3, 4, 4), torch.tensor([1, 5, 6, 7, 2]))]
    paths, images, cap, lens = R.create_batch(data)
    assert paths == ("b", "a") and lens == [5, 3]
    assert cap.dtype == torch.int64 and cap.tolist() == [[1, 5, 6, 7, 2], [1, 5, 2, 0, 0]]
    assert images.shape == (2, 3, 4, 4) and images[0, 0, 0, 0] == 1


def test_resnet_bad_version_raises():
    with pytest.raises(ValueError):
        R.resnet_conv_list(42)  # cnn.py:33
