"""Beam decoding on MI355X.

Two different things exist in the reference and both are mirrored:

* ``quirky_beam``  -- the live path of ``rnn.py:60-108`` (``sentence_index(beam_size>0)``, bs=1): one
  recurrent state threaded through every beam, candidates ranked by the current-step raw logit only.
* ``beam_search``  -- the textbook search of ``beam_search.py:45-97`` (cumulative -log p, <end> moves a
  node to the hypotheses, stable sort by cost), batched over images: every step is ONE set of device
  launches over the nodes of all images (st_embedding_rows, st_gather_state, st_rnn_step, st_softmax_topk and,
  by default, st_beam_select for the fringe selection, so that a search needs one host round trip in all).
  The reference's list bookkeeping (``Node`` parents, stable ``sorted``) is reproduced literally, including its
  float32 cost accumulation under NumPy 2 and the fact that nodes ending on the last iteration are never
  harvested; ``beam_search_host`` keeps that bookkeeping on the host per iteration (the cross-check).
"""
import ctypes as C

import numpy as np
import torch

from . import ops
from ._lib import ST_BF16, ST_F32, check, lib
from .rnn import CAP_MAX, _cp, _stream, up8


def _dtc(dt):
    return ST_F32 if dt == torch.float32 else ST_BF16


class _Stepper:
    """Device-side single-step decoder over n rows with explicit state tensors [L][n][H]."""

    def __init__(self, rnn):
        self.rnn = rnn
        self.prm, self.keep = rnn._c_params()
        self.dt = rnn.compute_dtype
        self.dev = rnn.linear.weight.device
        self.L, self.H, self.E, self.V = rnn.num_layers, rnn.hidden, rnn.embed_dim, rnn.vocab_size
        self.Vp = up8(self.V)
        self.lstm = rnn.cell != "gru"

    def embed(self, ids):
        ids = torch.as_tensor(ids, dtype=torch.long, device=self.dev).contiguous()
        x = torch.empty(ids.numel(), self.E, device=self.dev, dtype=self.dt)
        check(lib().st_embedding_rows(_cp(self.keep[0]), _cp(ids), _cp(x), ids.numel(), self.E, self.V, self.E, _dtc(self.dt), _stream()),
              "st_embedding_rows")
        return x

    def gather(self, state, idx):
        idx = torch.as_tensor(idx, dtype=torch.int32, device=self.dev).contiguous()
        out = []
        for s in state:
            d = torch.empty(self.L, idx.numel(), self.H, device=self.dev, dtype=self.dt)
            check(lib().st_gather_state(_cp(s), _cp(idx), _cp(d), self.L, s.shape[1], idx.numel(), self.H, _dtc(self.dt), _stream()),
                  "st_gather_state")
            out.append(d)
        return tuple(out)

    def step(self, x, state, want_logits=True):
        n = x.shape[0]
        h_out = torch.empty(self.L, n, self.H, device=self.dev, dtype=self.dt)
        c_out = torch.empty_like(h_out) if self.lstm else None
        logits = torch.empty(n, self.Vp, device=self.dev, dtype=torch.float32) if want_logits else None
        h_in = state[0] if state is not None else None
        c_in = state[1] if (state is not None and self.lstm) else None
        check(lib().st_rnn_step(C.byref(self.prm), _cp(x), n, _cp(h_in), _cp(c_in), _cp(h_out), _cp(c_out), _cp(logits), self.Vp,
                                _stream()), "st_rnn_step")
        return logits, ((h_out, c_out) if self.lstm else (h_out,))

    def topk(self, logits, k, raw):
        n = logits.shape[0]
        p = torch.empty(n, k, device=self.dev, dtype=torch.float32)
        i = torch.empty(n, k, device=self.dev, dtype=torch.long)
        check(lib().st_softmax_topk(_cp(logits), self.Vp, n, self.V, k, _cp(p), _cp(i), int(raw), _stream()), "st_softmax_topk")
        return p.cpu().numpy(), i.cpu().numpy()


def _feat(rnn, cnn_feature):
    f = cnn_feature.detach().contiguous()
    return f if f.dtype == rnn.compute_dtype else ops.cast(f.float(), rnn.compute_dtype)


def quirky_beam(rnn, cnn_feature, beam_size, steps=CAP_MAX):
    """rnn.py:60-108, bs=1 only (main.py:81-82 forces batch_size=1 when --beam_size>0)."""
    if cnn_feature.shape[0] != 1:
        raise ValueError("beam_size > 0 only works with batch_size=1 (rnn.py:60)")
    with torch.no_grad():
        st = _Stepper(rnn)
        logits, state = st.step(_feat(rnn, cnn_feature), None)                     # rnn.py:61-62
        _, ti = st.topk(logits, beam_size, raw=True)                               # rnn.py:63
        old_word = [int(w) for w in ti[0]]
        old_sent = [[w] for w in old_word]
        idx = 1
        while idx < steps:                                                         # rnn.py:78
            idx += 1
            new_sent, new_word, new_prob = [], [], []
            for k in range(beam_size):
                logits, state = st.step(st.embed([old_word[k]]), state)            # rnn.py:85-88: ONE shared state
                tv, ti = st.topk(logits, beam_size, raw=True)                      # rnn.py:90-91
                for j in range(beam_size):
                    new_sent.append(old_sent[k] + [int(ti[0, j])])
                    new_word.append(int(ti[0, j]))
                    new_prob.append(float(tv[0, j]))
            old_sent = [x for _, x in sorted(zip(new_prob, new_sent), reverse=True)][:beam_size]   # rnn.py:102
            old_word = [x for _, x in sorted(zip(new_prob, new_word), reverse=True)][:beam_size]   # rnn.py:103
        return torch.tensor(old_sent[0], dtype=torch.long, device=cnn_feature.device)   # rnn.py:106-108


def beam_search(rnn, cnn_feature, beam_width=4, num_hypotheses=1, max_length=50, start_id=1, end_id=2, on_device=True):
    """beam_search.py:45-97 for every image of the batch; see ``beam_search_host`` for the contract.  With ``on_device`` (and
    beam_width**2 <= 64) the fringe selection of every iteration also runs on the GPU (``st_beam_select``) over fixed
    (image, slot) rows, so the whole search needs ONE host round trip instead of three per iteration; the Node bookkeeping
    (harvest order, stable sort of the hypotheses, parent back-tracking) is replayed on the host from the recorded fringes.
    The only arithmetic that differs from the host path is -log p: a correctly rounded float32 logarithm on the device, NumPy's
    float32 log on the host (<= 1 ulp apart; the golden sequences are reproduced by both)."""
    k = min(beam_width, rnn.vocab_size)
    if not on_device or beam_width * k > 64:
        return beam_search_host(rnn, cnn_feature, beam_width, num_hypotheses, max_length, start_id, end_id)
    with torch.no_grad():
        st = _Stepper(rnn)
        dev = st.dev
        B, W = cnn_feature.shape[0], beam_width
        n = B * W
        _, state = st.step(_feat(rnn, cnn_feature), None, want_logits=False)      # state after the image-feature step (rnn.py:41,49)
        state = st.gather(state, torch.arange(B, device=dev, dtype=torch.int32).repeat_interleave(W))   # slot (b, w) <- image b
        tok = torch.zeros(B, W, device=dev, dtype=torch.long); tok[:, 0] = start_id    # beam_search.py:66
        cost = torch.full((B, W), float("inf"), device=dev, dtype=torch.float32); cost[:, 0] = 0.0
        done = torch.zeros(B, device=dev, dtype=torch.uint8)
        rec_tok = torch.zeros(max_length + 1, B, W, device=dev, dtype=torch.long)
        rec_cost = torch.full((max_length + 1, B, W), float("inf"), device=dev, dtype=torch.float32)
        rec_par = torch.full((max_length + 1, B, W), -1, device=dev, dtype=torch.int32)
        rec_end = torch.zeros(max_length, B, W, device=dev, dtype=torch.uint8)
        rec_tok[0], rec_cost[0] = tok, cost
        gidx = torch.empty(n, device=dev, dtype=torch.int32)
        tp = torch.empty(n, k, device=dev, dtype=torch.float32)
        ti = torch.empty(n, k, device=dev, dtype=torch.long)
        for t in range(max_length):                                                   # beam_search.py:69
            x = st.embed(tok.view(-1))
            logits, new_state = st.step(x, state)
            check(lib().st_softmax_topk(_cp(logits), st.Vp, n, st.V, k, _cp(tp), _cp(ti), 0, _stream()), "st_softmax_topk")
            ntok, ncost = rec_tok[t + 1], rec_cost[t + 1]
            check(lib().st_beam_select(_cp(tok), _cp(cost), _cp(done), _cp(tp), _cp(ti), B, W, k, end_id, _cp(ntok), _cp(ncost),
                                       _cp(rec_par[t + 1]), _cp(rec_end[t]), _cp(gidx), _stream()), "st_beam_select")
            state = st.gather(new_state, gidx)
            tok, cost = ntok, ncost
        h_tok, h_cost, h_par, h_end = rec_tok.cpu().numpy(), rec_cost.cpu().numpy(), rec_par.cpu().numpy(), rec_end.cpu().numpy()
    out = []
    for b in range(B):
        hyp = []                                                                      # (iteration of the node, slot, cost) in harvest order
        for t, w in zip(*np.nonzero(h_end[:, b])):                                    # iteration-major, slot order inside (beam_search.py:72-76)
            hyp.append((int(t), int(w), h_cost[t, b, w]))
        res = []
        for t, w, c in sorted(hyp, key=lambda e: e[2])[:num_hypotheses]:              # beam_search.py:96 (stable)
            seq = []
            while t >= 0 and w >= 0:
                seq.append(int(h_tok[t, b, w]))
                w = int(h_par[t, b, w]) if t > 0 else -1
                t -= 1
            res.append((seq[::-1], float(c)))
        out.append(res)
    return out


def beam_search_host(rnn, cnn_feature, beam_width=4, num_hypotheses=1, max_length=50, start_id=1, end_id=2):
    """beam_search.py:45-97 for every image of the batch.  Returns, per image, a list of at most
    `num_hypotheses` (token_sequence, cum_cost) pairs; the list is empty when no beam ever emitted
    `end_id` in time (the reference returns [] then).

    Bookkeeping is vectorised over the batch (fixed (B, W, k) candidate blocks; the reference's per-image lists of
    Node objects were 95 % of the time at 256 images) and reproduces the reference's order exactly: candidates are
    laid out node-major in fringe order, each node's k successors in ASCENDING probability (argsort(p)[-k:]), and cut
    with a STABLE sort on the float32 cumulative cost; finished nodes are harvested at the start of an iteration (so
    nodes that end on the last iteration are never harvested), hypotheses are stably sorted by cost at the end."""
    with torch.no_grad():
        st = _Stepper(rnn)
        B = cnn_feature.shape[0]
        W = beam_width
        k = min(beam_width, st.V)
        _, state = st.step(_feat(rnn, cnn_feature), None, want_logits=False)      # state after the image-feature step (rnn.py:41,49)
        # global node table (parent pointers) for the final back-tracking
        node_parent = [np.full(B, -1, dtype=np.int64)]
        node_value = [np.full(B, start_id, dtype=np.int64)]
        n_nodes = B
        tok = np.full((B, W), -1, dtype=np.int64); tok[:, 0] = start_id              # beam_search.py:66
        cost = np.full((B, W), np.inf, dtype=np.float32); cost[:, 0] = 0.0
        row = np.zeros((B, W), dtype=np.int64); row[:, 0] = np.arange(B)             # row of the node's state
        nid = np.full((B, W), -1, dtype=np.int64); nid[:, 0] = np.arange(B)
        valid = np.zeros((B, W), dtype=bool); valid[:, 0] = True
        done = np.zeros(B, dtype=bool)
        hyp_nodes = [[] for _ in range(B)]                                            # (node id, cost) in harvest order
        for _ in range(max_length):                                                   # beam_search.py:69
            ended = valid & (tok == end_id) & ~done[:, None]                          # beam_search.py:72-76
            if ended.any():
                for b, w in zip(*np.nonzero(ended)):
                    hyp_nodes[b].append((int(nid[b, w]), cost[b, w]))
            live = valid & ~ended & ~done[:, None]
            done |= ~live.any(axis=1)                                                 # beam_search.py:78-79 (break)
            live &= ~done[:, None]
            lb, lw = np.nonzero(live)                                                 # image-major, fringe order inside an image
            if lb.size == 0:
                break
            x = st.embed(tok[lb, lw])
            logits, new_state = st.step(x, st.gather(state, row[lb, lw]))
            tp, ti = st.topk(logits, k, raw=False)                                    # descending; argsort(p)[-k:] is ascending
            nll = -np.log(tp[:, ::-1].astype(np.float32))                             # beam_search.py:87-92
            cand_cost = np.full((B, W, k), np.inf, dtype=np.float32)
            cand_tok = np.zeros((B, W, k), dtype=np.int64)
            cand_row = np.zeros((B, W, k), dtype=np.int64)
            cand_par = np.zeros((B, W, k), dtype=np.int64)
            cand_cost[lb, lw] = cost[lb, lw][:, None] + nll                           # float32 + float32 (NEP 50: the root's 0.0 is weak)
            cand_tok[lb, lw] = ti[:, ::-1]
            cand_row[lb, lw] = np.arange(lb.size)[:, None]
            cand_par[lb, lw] = nid[lb, lw][:, None]
            flat = cand_cost.reshape(B, W * k)
            order = np.argsort(flat, axis=1, kind="stable")[:, :W]                    # beam_search.py:94 (stable sorted()[:beam_width])
            take = np.take_along_axis
            new_cost = take(flat, order, 1)
            valid = np.isfinite(new_cost) & ~done[:, None]
            tok = take(cand_tok.reshape(B, W * k), order, 1)
            row = take(cand_row.reshape(B, W * k), order, 1)
            par = take(cand_par.reshape(B, W * k), order, 1)
            cost = np.where(valid, new_cost, np.float32(np.inf)).astype(np.float32)
            # register the new nodes
            vb, vw = np.nonzero(valid)
            nid = np.full((B, W), -1, dtype=np.int64)
            nid[vb, vw] = n_nodes + np.arange(vb.size)
            node_parent.append(par[vb, vw]); node_value.append(tok[vb, vw])
            n_nodes += vb.size
            state = new_state
        parent = np.concatenate(node_parent); value = np.concatenate(node_value)
        out = []
        for b in range(B):
            hy = sorted(hyp_nodes[b], key=lambda t: t[1])[:num_hypotheses]            # beam_search.py:96 (stable)
            res = []
            for node, c in hy:
                seq = []
                while node >= 0:
                    seq.append(int(value[node])); node = int(parent[node])
                res.append((seq[::-1], float(c)))
            out.append(res)
        return out
