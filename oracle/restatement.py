"""CPU restatement of the show-tell hot path (TEST INFRASTRUCTURE ONLY).

This file is the parity oracle for the MI355X build.  It restates, with plain
``torch`` CPU fp32 tensor arithmetic (explicit gate equations, explicit packed
sequence bookkeeping, explicit attention), what the reference computes through
``torch.nn.GRU/LSTM/Linear/Embedding/BatchNorm`` and torchvision's ResNet.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it.  The product path (``show-tell_amd/``) never does.

Pinning
-------
* Decoders / loss / beam / BLEU: pinned against the reference's own classes,
  imported in the authoring container by ``oracle/gen_golden.py`` (which writes
  ``tests/golden/*.npz``) and checked live by ``tests/test_oracle_pin.py`` when
  ``/root/reference`` is present.
* Encoder backbone: the arithmetic lives in torchvision 0.3.0
  (``models.resnet*``; reference call sites cnn.py:23-34, cnn_attn.py:23-34),
  which is absent from /root/reference and not installed; the reference holds
  no fixture for it.  The backbone is restated from the published ResNet v1.5
  definition and is **parity unpinned** at that boundary; the head
  (cnn.py:37-51) is pinned through torch ops.

All functions take a flat ``params`` dict keyed exactly like the reference
modules' ``state_dict()``.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

CAP_MAX = 25  # rnn.py:39, rnn_attn.py:53

# --------------------------------------------------------------------------
# Packed-sequence bookkeeping (torch.nn.utils.rnn.pack_padded_sequence as used
# at rnn.py:31 and main.py:145; inputs are length-sorted desc, utils.py:66)
# --------------------------------------------------------------------------

def batch_sizes(lens):
    """batch_sizes[t] = #{b : lens[b] > t}; lens must be sorted descending."""
    lens = [int(l) for l in lens]
    assert all(lens[i] >= lens[i + 1] for i in range(len(lens) - 1)), "lens must be sorted desc"
    return [sum(1 for l in lens if l > t) for t in range(lens[0])]


def pack_rows(x, lens):
    """(B,T,...) -> time-major packed rows (N_tok,...); main.py:145, rnn_attn.py:115."""
    bs = batch_sizes(lens)
    return torch.cat([x[:b, t] for t, b in enumerate(bs)], 0)


# --------------------------------------------------------------------------
# Cells (torch.nn.GRU / LSTM semantics, gate order [r,z,n] / [i,f,g,o])
# --------------------------------------------------------------------------

def gru_cell(x, h, w_ih, w_hh, b_ih, b_hh):
    H = h.shape[1]
    gx = x @ w_ih.t() + b_ih
    gh = h @ w_hh.t() + b_hh
    r = torch.sigmoid(gx[:, :H] + gh[:, :H])
    z = torch.sigmoid(gx[:, H:2 * H] + gh[:, H:2 * H])
    n = torch.tanh(gx[:, 2 * H:] + r * gh[:, 2 * H:])
    return (1.0 - z) * n + z * h


def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    H = h.shape[1]
    g = x @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
    i = torch.sigmoid(g[:, :H])
    f = torch.sigmoid(g[:, H:2 * H])
    gg = torch.tanh(g[:, 2 * H:3 * H])
    o = torch.sigmoid(g[:, 3 * H:])
    c2 = f * c + i * gg
    return o * torch.tanh(c2), c2


def _layer_w(params, l, prefix="unit."):
    return (params[f"{prefix}weight_ih_l{l}"], params[f"{prefix}weight_hh_l{l}"],
            params[f"{prefix}bias_ih_l{l}"], params[f"{prefix}bias_hh_l{l}"])


def num_layers_of(params, prefix="unit."):
    l = 0
    while f"{prefix}weight_ih_l{l}" in params:
        l += 1
    return l


def rnn_step(params, x, h, c=None, cell="gru"):
    """One time step through all layers.  h (and c): (L,B,H).  Returns (top, h', c')."""
    L = num_layers_of(params)
    hs, cs = [], []
    inp = x
    for l in range(L):
        w = _layer_w(params, l)
        if cell == "gru":
            hl = gru_cell(inp, h[l], *w)
        else:
            hl, cl = lstm_cell(inp, h[l], c[l], *w)
            cs.append(cl)
        hs.append(hl)
        inp = hl
    return inp, torch.stack(hs, 0), (torch.stack(cs, 0) if cell != "gru" else None)


def rnn_packed(params, x_btE, lens, cell="gru"):
    """Multi-layer RNN over a packed sequence, zero initial state.

    x_btE: (B,T,in) batch-first, rows beyond lens ignored.  Returns the top
    layer's packed output rows (N_tok,H), time-major (nn.GRU on PackedSequence,
    rnn.py:32 / rnn_lstm.py:30).
    """
    bs = batch_sizes(lens)
    B = x_btE.shape[0]
    L = num_layers_of(params)
    H = params["unit.weight_hh_l0"].shape[1]
    seq = [x_btE[:b, t] for t, b in enumerate(bs)]
    for l in range(L):
        w = _layer_w(params, l)
        h = x_btE.new_zeros(B, H)
        c = x_btE.new_zeros(B, H)
        out = []
        for t, b in enumerate(bs):
            if cell == "gru":
                hn = gru_cell(seq[t], h[:b], *w)
            else:
                hn, cn = lstm_cell(seq[t], h[:b], c[:b], *w)
                c = torch.cat([cn, c[b:]], 0)
            h = torch.cat([hn, h[b:]], 0)
            out.append(hn)
        seq = out
    return torch.cat(seq, 0)


# --------------------------------------------------------------------------
# A5 / A14: teacher-forced decoder forward (rnn.py:27-35, LSTM/rnn_lstm.py:25-33)
# --------------------------------------------------------------------------

def rnn_forward(params, cnn_feature, image_caption, caption_size, cell="gru"):
    emb = params["embeddings.weight"][image_caption]                      # rnn.py:29
    raw = torch.cat((cnn_feature.unsqueeze(1), emb), 1)                   # rnn.py:30
    top = rnn_packed(params, raw, caption_size, cell)                     # rnn.py:31-32
    return top @ params["linear.weight"].t() + params["linear.bias"]      # rnn.py:33


def ce_loss(logits, target):
    """nn.CrossEntropyLoss() default: mean over rows (main.py:94,149)."""
    return F.cross_entropy(logits, target)


def gru_train_loss(params, cnn_feature, image_caption, caption_size, cell="gru"):
    """main.py:145-149: target = packed caption, CE mean over N_tok."""
    target = pack_rows(image_caption, caption_size)
    logits = rnn_forward(params, cnn_feature, image_caption, caption_size, cell)
    return ce_loss(logits, target), logits, target


# --------------------------------------------------------------------------
# A6: greedy decode, exactly 25 steps (rnn.py:37-58, rnn_lstm.py:35-57)
# --------------------------------------------------------------------------

def rnn_greedy(params, cnn_feature, cell="gru", steps=CAP_MAX, return_logits=False):
    B = cnn_feature.shape[0]
    L = num_layers_of(params)
    H = params["unit.weight_hh_l0"].shape[1]
    h = cnn_feature.new_zeros(L, B, H)
    c = cnn_feature.new_zeros(L, B, H) if cell != "gru" else None
    x = cnn_feature
    ids, all_logits = [], []
    for _ in range(steps):
        top, h, c = rnn_step(params, x, h, c, cell)
        logits = top @ params["linear.weight"].t() + params["linear.bias"]
        tok = logits.max(1)[1]                                            # rnn.py:51
        ids.append(tok)
        all_logits.append(logits)
        x = params["embeddings.weight"][tok]                              # rnn.py:53
    out = torch.stack(ids, 1).squeeze()                                   # rnn.py:56
    if return_logits:
        return out, torch.stack(all_logits, 1)
    return out


# --------------------------------------------------------------------------
# A7: the live "beam" of rnn.py:60-108 (bs=1; shared hidden state threaded
# through every beam; ranking by current-step raw logit only)
# --------------------------------------------------------------------------

def rnn_beam_quirky(params, cnn_feature, beam_size, steps=CAP_MAX):
    assert cnn_feature.shape[0] == 1, "rnn.py:60 only works with batch_size=1"
    L = num_layers_of(params)
    H = params["unit.weight_hh_l0"].shape[1]
    W, b = params["linear.weight"], params["linear.bias"]
    emb = params["embeddings.weight"]
    h = cnn_feature.new_zeros(L, 1, H)
    top, h, _ = rnn_step(params, cnn_feature, h)                          # rnn.py:61
    logits = top @ W.t() + b
    topk = logits.topk(k=beam_size, dim=1)[1]                             # rnn.py:63
    old_word = [int(topk[0, k]) for k in range(beam_size)]
    old_sent = [[w] for w in old_word]
    idx = 1
    while idx < steps:                                                    # rnn.py:78
        idx += 1
        new_sent, new_word, new_prob = [], [], []
        for k in range(beam_size):
            x = emb[torch.tensor([old_word[k]])]
            top, h, _ = rnn_step(params, x, h)                            # rnn.py:87 (shared h)
            logits = top @ W.t() + b
            tv, ti = logits.topk(k=beam_size, dim=1)                      # rnn.py:90-91
            for j in range(beam_size):
                new_sent.append(old_sent[k] + [int(ti[0, j])])
                new_word.append(int(ti[0, j]))
                new_prob.append(float(tv[0, j]))
        # rnn.py:102-103: two independent sorts on (prob, payload) tuples, descending
        old_sent = [x for _, x in sorted(zip(new_prob, new_sent), reverse=True)][:beam_size]
        old_word = [x for _, x in sorted(zip(new_prob, new_word), reverse=True)][:beam_size]
    return torch.tensor(old_sent[0], dtype=torch.long)                    # rnn.py:106-108


# --------------------------------------------------------------------------
# A8-A10: soft attention decoder (Attention/rnn_attn.py, rnn_attn_LSTM.py)
# --------------------------------------------------------------------------

def attention_net(params, img_feat, hidden_state):
    """rnn_attn.py:21-31.  img_feat (B,P,F), hidden_state (B,H) -> (z (B,F), alpha (B,P))."""
    att1 = img_feat @ params["attn.encoder_att.weight"].t() + params["attn.encoder_att.bias"]
    att2 = hidden_state @ params["attn.decoder_att.weight"].t() + params["attn.decoder_att.bias"]
    e = F.leaky_relu(att1 + att2.unsqueeze(1), 0.2)
    att = (e @ params["attn.full_att.weight"].t() + params["attn.full_att.bias"]).squeeze(2)
    alpha = torch.softmax(att, dim=1)
    z = (img_feat * alpha.unsqueeze(2)).sum(dim=1)
    return z, alpha


def _attn_init(params, cnn_feature, cell):
    L = num_layers_of(params)
    mean = cnn_feature.mean(dim=2)                                        # rnn_attn.py:62
    h0 = mean @ params["init_h.weight"].t() + params["init_h.bias"]
    h = h0.unsqueeze(0).repeat(L, 1, 1)
    c = None
    if cell != "gru":
        c0 = mean @ params["init_c.weight"].t() + params["init_c.bias"]   # rnn_attn_LSTM.py:63
        c = c0.unsqueeze(0).repeat(L, 1, 1)
    return h, c


def attn_forward(params, cnn_feature, image_caption, caption_size, cell="gru"):
    """rnn_attn.py:98-118 (train branch 63-76).  cnn_feature (B,F,P).

    Returns (packed logits (N_tok,V), alphas (B,T,P) zero beyond each length).
    Quirk kept: the input token at step t is caption[:,t] (rnn_attn.py:70).
    """
    B, T = image_caption.shape
    V = params["linear.weight"].shape[0]
    P = cnn_feature.shape[2]
    emb = params["embeddings.weight"][image_caption]
    h, c = _attn_init(params, cnn_feature, cell)
    feat_bpf = cnn_feature.transpose(1, 2)
    preds = cnn_feature.new_zeros(B, T, V)
    alphas = cnn_feature.new_zeros(B, T, P)
    for t in range(T):
        bt = sum(1 for l in caption_size if l > t)                        # rnn_attn.py:68
        if bt == 0:
            break
        z, alpha = attention_net(params, feat_bpf[:bt], h[-1, :bt])
        ez = z @ params["embed.weight"].t() + params["embed.bias"]
        x = torch.cat([emb[:bt, t], ez], 1)
        top, hn, cn = rnn_step(params, x, h[:, :bt], None if c is None else c[:, :bt], cell)
        h = hn  # batch shrinks monotonically; rows >= bt are never used again
        c = cn
        preds[:bt, t] = top @ params["linear.weight"].t() + params["linear.bias"]
        alphas[:bt, t] = alpha
    return pack_rows(preds, caption_size), alphas


def attn_train_loss(params, cnn_feature, image_caption, caption_size, alpha_c=1.0, cell="gru"):
    """Attention/main_attn.py:126-131."""
    target = pack_rows(image_caption, caption_size)
    logits, alphas = attn_forward(params, cnn_feature, image_caption, caption_size, cell)
    loss = ce_loss(logits, target)
    loss = loss + alpha_c * ((1.0 - alphas.sum(dim=1)) ** 2).mean()
    return loss, logits, alphas


def attn_greedy(params, cnn_feature, start_id=1, cell="gru", steps=CAP_MAX):
    """rnn_attn.py:120-145 with the test branch 77-94."""
    B = cnn_feature.shape[0]
    h, c = _attn_init(params, cnn_feature, cell)
    feat_bpf = cnn_feature.transpose(1, 2)
    tok_emb = params["embeddings.weight"][torch.full((B,), start_id, dtype=torch.long)]
    ids = []
    for _ in range(steps):
        z, _ = attention_net(params, feat_bpf, h[-1])
        ez = z @ params["embed.weight"].t() + params["embed.bias"]
        top, h, c = rnn_step(params, torch.cat([tok_emb, ez], 1), h, c, cell)
        logits = top @ params["linear.weight"].t() + params["linear.bias"]
        tok = logits.max(1)[1]
        ids.append(tok)
        tok_emb = params["embeddings.weight"][tok]
    return torch.stack(ids, 1).squeeze()


# --------------------------------------------------------------------------
# A13: beam_search.py:18-97 (numpy; float32 probs -> float64 costs)
# --------------------------------------------------------------------------

class Node:
    def __init__(self, parent, state, value, cost, extras):
        self.value = value
        self.parent = parent
        self.state = state.flatten() if state is not None else None
        self.cum_cost = parent.cum_cost + cost if parent else cost
        self.length = 1 if parent is None else parent.length + 1
        self.extras = extras

    def to_sequence_of_values(self):
        seq, n = [], self
        while n:
            seq.insert(0, n.value)
            n = n.parent
        return seq


def beam_search(initial_state_function, generate_function, X, start_id, end_id,
                beam_width=4, num_hypotheses=1, max_length=50):
    if isinstance(X, list) or X.ndim == 1:
        X = np.array([X], dtype=np.int32).T
    next_fringe = [Node(None, initial_state_function(X), start_id, 0.0, None)]
    hypotheses = []
    for _ in range(max_length):
        fringe = []
        for n in next_fringe:
            (hypotheses if n.value == end_id else fringe).append(n)
        if not fringe:
            break
        Y_tm1 = np.array([n.value for n in fringe], dtype=np.int32)
        state_tm1 = np.array([n.state for n in fringe], dtype=np.float32)
        state_t, p_t, extras_t = generate_function(X, Y_tm1, state_tm1)
        Y_t = np.argsort(p_t, axis=1)[:, -beam_width:]                    # beam_search.py:84
        next_fringe = []
        for Y_t_n, p_t_n, extras_t_n, state_t_n, n in zip(Y_t, p_t, extras_t, state_t, fringe):
            nll = -np.log(p_t_n[Y_t_n])
            for y, c in zip(Y_t_n, nll):
                next_fringe.append(Node(n, state_t_n, y, c, extras_t_n))
        next_fringe = sorted(next_fringe, key=lambda n: n.cum_cost)[:beam_width]
    hypotheses.sort(key=lambda n: n.cum_cost)
    return hypotheses[:num_hypotheses]


def gru_beam_callbacks(params, cnn_feature_row):
    """Callbacks that drive beam_search with the GRU captioner for ONE image.

    The captioner has no <start> input: step 0 feeds the image feature
    (rnn.py:41), so the initial state returned here is the state AFTER the
    feature step and the root node carries start_id as a placeholder value.
    State is the flattened (L,H) hidden (beam_search.py:23 flattens).
    """
    L = num_layers_of(params)
    H = params["unit.weight_hh_l0"].shape[1]

    def initial_state(_X):
        h = cnn_feature_row.new_zeros(L, 1, H)
        _, h, _ = rnn_step(params, cnn_feature_row.view(1, -1), h)
        return h.detach().numpy().astype(np.float32)

    def generate(_X, Y_tm1, state_tm1):
        n = len(Y_tm1)
        h = torch.from_numpy(state_tm1).view(n, L, H).transpose(0, 1).contiguous()
        x = params["embeddings.weight"][torch.from_numpy(Y_tm1.astype(np.int64))]
        top, h2, _ = rnn_step(params, x, h)
        logits = top @ params["linear.weight"].t() + params["linear.bias"]
        p = torch.softmax(logits, 1).detach().numpy().astype(np.float32)
        st = h2.transpose(0, 1).contiguous().view(n, L * H).detach().numpy().astype(np.float32)
        return st, p, [None] * n

    return initial_state, generate


# --------------------------------------------------------------------------
# A1-A3: encoder.  Backbone restated from the published torchvision ResNet
# v1.5 definition (stride on the 3x3; BN eps 1e-5, momentum 0.1).
# --------------------------------------------------------------------------

RESNET_SPECS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]),
                50: ("bottleneck", [3, 4, 6, 3]), 101: ("bottleneck", [3, 4, 23, 3]),
                152: ("bottleneck", [3, 8, 36, 3])}


def resnet_conv_list(version):
    """[(key_prefix, cin, cout, k, stride, pad, bn_prefix)] in forward order."""
    if version not in RESNET_SPECS:
        raise ValueError("Please specify a valid ResNet version. %d doesn't exist." % version)  # cnn.py:33
    kind, blocks = RESNET_SPECS[version]
    exp = 4 if kind == "bottleneck" else 1
    out = [("model.0", 3, 64, 7, 2, 3, "model.1")]
    inpl = 64
    for li, (planes, nb) in enumerate(zip([64, 128, 256, 512], blocks)):
        for bi in range(nb):
            s = (1 if li == 0 else 2) if bi == 0 else 1
            p = f"model.{4 + li}.{bi}"
            if kind == "bottleneck":
                out.append((p + ".conv1", inpl, planes, 1, 1, 0, p + ".bn1"))
                out.append((p + ".conv2", planes, planes, 3, s, 1, p + ".bn2"))
                out.append((p + ".conv3", planes, planes * 4, 1, 1, 0, p + ".bn3"))
            else:
                out.append((p + ".conv1", inpl, planes, 3, s, 1, p + ".bn1"))
                out.append((p + ".conv2", planes, planes, 3, 1, 1, p + ".bn2"))
            if bi == 0 and (s != 1 or inpl != planes * exp):
                out.append((p + ".downsample.0", inpl, planes * exp, 1, s, 0, p + ".downsample.1"))
            inpl = planes * exp
    return out


def init_encoder_params(version=101, embed_dim=256, seed=1, attn=False):
    """Random-init parameters with torchvision-style inits (no pretrained weights offline)."""
    g = torch.Generator().manual_seed(seed)
    p = OrderedDict()
    feat = 64
    for key, cin, cout, k, s, pad, bn in resnet_conv_list(version):
        std = math.sqrt(2.0 / (cout * k * k))  # kaiming_normal_, fan_out, relu
        p[key + ".weight"] = torch.randn(cout, cin, k, k, generator=g) * std
        p[bn + ".weight"] = torch.ones(cout)
        p[bn + ".bias"] = torch.zeros(cout)
        p[bn + ".running_mean"] = torch.zeros(cout)
        p[bn + ".running_var"] = torch.ones(cout)
        p[bn + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
        feat = cout if "downsample" not in key else feat
    kind, _ = RESNET_SPECS[version]
    F_in = 2048 if kind == "bottleneck" else 512
    bound = 1.0 / math.sqrt(F_in)
    p["linear_secondlast_layer.weight"] = torch.randn(embed_dim, F_in, generator=g) * 0.05  # cnn.py:41
    p["linear_secondlast_layer.bias"] = (torch.rand(embed_dim, generator=g) * 2 - 1) * bound
    p["last_layer.weight"] = torch.ones(embed_dim)
    p["last_layer.bias"] = torch.zeros(embed_dim)                                            # cnn.py:42
    p["last_layer.running_mean"] = torch.zeros(embed_dim)
    p["last_layer.running_var"] = torch.ones(embed_dim)
    p["last_layer.num_batches_tracked"] = torch.tensor(0, dtype=torch.long)
    return p


def _bn2d(params, x, bn, train, momentum=0.1, eps=1e-5):
    rm, rv = params[bn + ".running_mean"], params[bn + ".running_var"]
    y = F.batch_norm(x, rm, rv, params[bn + ".weight"], params[bn + ".bias"], train, momentum, eps)
    if train:
        params[bn + ".num_batches_tracked"] += 1
    return y


def block_forward(params, x, version, li, bi, train=False):
    """ONE residual block of torchvision's ResNet (BasicBlock / Bottleneck.forward, v1.5: stride on the 3x3), block `bi` of
    layer{li + 1}: the body of backbone_forward's loop, callable on its own so that a kernel test can feed it the HIP path's
    own output of the previous block (block-by-block parity: no cross-block amplification of storage rounding)."""
    kind, _ = RESNET_SPECS[version]
    p = f"model.{4 + li}.{bi}"
    s = (1 if li == 0 else 2) if bi == 0 else 1
    idt = x
    if kind == "bottleneck":
        o = F.relu(_bn2d(params, F.conv2d(x, params[p + ".conv1.weight"]), p + ".bn1", train))
        o = F.relu(_bn2d(params, F.conv2d(o, params[p + ".conv2.weight"], None, s, 1), p + ".bn2", train))
        o = _bn2d(params, F.conv2d(o, params[p + ".conv3.weight"]), p + ".bn3", train)
    else:
        o = F.relu(_bn2d(params, F.conv2d(x, params[p + ".conv1.weight"], None, s, 1), p + ".bn1", train))
        o = _bn2d(params, F.conv2d(o, params[p + ".conv2.weight"], None, 1, 1), p + ".bn2", train)
    if p + ".downsample.0.weight" in params:
        idt = _bn2d(params, F.conv2d(x, params[p + ".downsample.0.weight"], None, s), p + ".downsample.1", train)
    return F.relu(o + idt)


def backbone_forward(params, x, version=101, train=False, avgpool=True, taps=None):
    """torchvision resnet children()[:-1] (cnn.py:34) or [:-2] (cnn_attn.py:34).

    train=True puts every BatchNorm2d in batch-statistics mode and updates the
    running buffers in ``params`` in place (main.py:125 calls cnn.train()).
    ``taps`` (dict) optionally collects intermediate activations for kernel tests.
    """
    kind, blocks = RESNET_SPECS[version]
    x = F.conv2d(x, params["model.0.weight"], None, 2, 3)
    x = F.relu(_bn2d(params, x, "model.1", train))
    if taps is not None:
        taps["stem"] = x
    x = F.max_pool2d(x, 3, 2, 1)
    if taps is not None:
        taps["pool"] = x
    for li, nb in enumerate(blocks):
        for bi in range(nb):
            x = block_forward(params, x, version, li, bi, train)
            if taps is not None:
                taps[f"model.{4 + li}.{bi}"] = x
    if avgpool:
        x = F.adaptive_avg_pool2d(x, 1)
    return x


def encoder_forward(params, x, version=101, train=False):
    """cnn.py:44-51: backbone -> detach -> flatten -> Linear -> BatchNorm1d(momentum=0.01)."""
    f = backbone_forward(params, x, version, train, avgpool=True).detach()
    f = f.view(f.size(0), -1)
    y = f @ params["linear_secondlast_layer.weight"].t() + params["linear_secondlast_layer.bias"]
    y = F.batch_norm(y, params["last_layer.running_mean"], params["last_layer.running_var"],
                     params["last_layer.weight"], params["last_layer.bias"], train, 0.01, 1e-5)
    if train:
        params["last_layer.num_batches_tracked"] += 1
    return y


def encoder_attn_forward(params, x, version=101, train=False):
    """cnn_attn.py:44-52: backbone without avgpool -> detach -> (B,2048,49)."""
    f = backbone_forward(params, x, version, train, avgpool=False).detach()
    return f.view(f.size(0), f.size(1), -1)


# --------------------------------------------------------------------------
# Decoder parameter init (torch defaults of nn.Embedding / nn.GRU / nn.Linear)
# --------------------------------------------------------------------------

def init_decoder_params(E, H, V, L, cell="gru", seed=1, attn=None):
    """attn = None or dict(F=nos_filters, A=attention_dim) for RNN_Attn."""
    g = torch.Generator().manual_seed(seed)
    G = 3 if cell == "gru" else 4
    p = OrderedDict()
    p["embeddings.weight"] = torch.randn(V, E, generator=g)
    k = 1.0 / math.sqrt(H)
    u = lambda *s, b=k: (torch.rand(*s, generator=g) * 2 - 1) * b
    in0 = 2 * E if attn else E
    for l in range(L):
        p[f"unit.weight_ih_l{l}"] = u(G * H, in0 if l == 0 else H)
        p[f"unit.weight_hh_l{l}"] = u(G * H, H)
        p[f"unit.bias_ih_l{l}"] = u(G * H)
        p[f"unit.bias_hh_l{l}"] = u(G * H)
    p["linear.weight"] = u(V, H)
    p["linear.bias"] = u(V)
    if attn:
        Fd, A = attn["F"], attn["A"]
        kf, ka, kh = 1 / math.sqrt(Fd), 1 / math.sqrt(A), 1 / math.sqrt(H)
        p["init_h.weight"] = u(H, Fd, b=kf); p["init_h.bias"] = u(H, b=kf)
        if cell != "gru":
            p["init_c.weight"] = u(H, Fd, b=kf); p["init_c.bias"] = u(H, b=kf)
        p["attn.encoder_att.weight"] = u(A, Fd, b=kf); p["attn.encoder_att.bias"] = u(A, b=kf)
        p["attn.decoder_att.weight"] = u(A, H, b=kh); p["attn.decoder_att.bias"] = u(A, b=kh)
        p["attn.full_att.weight"] = u(1, A, b=ka); p["attn.full_att.bias"] = u(1, b=ka)
        p["embed.weight"] = u(E, Fd, b=kf); p["embed.bias"] = u(E, b=kf)
    return p


# --------------------------------------------------------------------------
# A15 + synthetic workload: the input contract of utils.create_batch
# --------------------------------------------------------------------------

def synthetic_captions(B, V, seed=1, mean=12.5, std=2.5, lo=6, hi=25):
    """Length-sorted (desc) zero-padded captions [1]+randint(4,V)+[2]; utils.py:61-77 layout."""
    rng = np.random.RandomState(seed)
    lens = np.clip(np.rint(rng.normal(mean, std, size=B)), lo, hi).astype(np.int64)
    lens = np.sort(lens)[::-1].copy()
    cap = np.zeros((B, int(lens[0])), dtype=np.int64)
    for b, l in enumerate(lens):
        cap[b, 0] = 1
        cap[b, 1:l - 1] = rng.randint(4, V, size=l - 2)
        cap[b, l - 1] = 2
    return torch.from_numpy(cap), [int(l) for l in lens]


def create_batch(data):
    """utils.py:61-77 on (path, image, caption_tensor) triples."""
    data = sorted(data, key=lambda x: len(x[2]), reverse=True)
    paths, images, captions = zip(*data)
    images = torch.stack(images, 0)
    lens = [len(c) for c in captions]
    tgt = torch.zeros(len(captions), max(lens)).long()
    for i, c in enumerate(captions):
        tgt[i, :lens[i]] = c[:lens[i]]
    return paths, images, tgt, lens


# --------------------------------------------------------------------------
# A11: one GRU-model training step (main.py:94-100, 136-152), oracle form
# --------------------------------------------------------------------------

def sgd_momentum_step(p, g, buf, lr, momentum):
    """torch.optim.SGD(lr, momentum), dampening 0, no nesterov/weight decay."""
    if buf is None:
        buf = g.clone()
    else:
        buf.mul_(momentum).add_(g)
    p.add_(buf, alpha=-lr)
    return buf


def adam_step(p, g, m, v, step, lr, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam(lr) defaults."""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------
# BLEU (evaluation/evaluation_metrics.py:117-317), 'closest' reflen option
# --------------------------------------------------------------------------

def _ngrams(words, n=4):
    c = {}
    for k in range(1, n + 1):
        for i in range(len(words) - k + 1):
            ng = tuple(words[i:i + k])
            c[ng] = c.get(ng, 0) + 1
    return len(words), c


def bleu_corpus(gts, res, n=4):
    """gts/res: dict id -> list of sentences (strings); res lists have one entry.
    Returns [BLEU-1..n] exactly as Bleu.compute_score(...)[0] (evaluation_metrics.py:293-314)."""
    small, tiny = 1e-9, 1e-15
    tot_guess, tot_correct = [0] * n, [0] * n
    testlen_sum, reflen_sum = 0, 0.0
    for key in gts.keys():
        refs = [r.split() for r in gts[key]]
        test = res[key][0].split()
        reflens, maxc = [], {}
        for r in refs:
            rl, cnt = _ngrams(r, n)
            reflens.append(rl)
            for ng, c in cnt.items():
                maxc[ng] = max(maxc.get(ng, 0), c)
        tl, cnt = _ngrams(test, n)
        reflen = min((abs(l - tl), l) for l in reflens)[1]
        testlen_sum += tl
        reflen_sum += reflen
        for k in range(n):
            tot_guess[k] += max(0, tl - k)
        for ng, c in cnt.items():
            tot_correct[len(ng) - 1] += min(maxc.get(ng, 0), c)
    bleus, bleu = [], 1.0
    for k in range(n):
        bleu *= float(tot_correct[k] + tiny) / (tot_guess[k] + small)
        bleus.append(bleu ** (1.0 / (k + 1)))
    ratio = (testlen_sum + tiny) / (reflen_sum + small)
    if ratio < 1:
        bleus = [b * math.exp(1 - 1 / ratio) for b in bleus]
    return bleus
