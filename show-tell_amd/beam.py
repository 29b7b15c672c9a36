"""Beam decoding on MI355X.

Two different things exist in the reference and both are mirrored:

* ``quirky_beam``  -- the live path of ``rnn.py:60-108`` (``sentence_index(beam_size>0)``, bs=1): one
  recurrent state threaded through every beam, candidates ranked by the current-step raw logit only.
* ``beam_search``  -- the textbook search of ``beam_search.py:45-97`` (cumulative -log p, <end> moves a
  node to the hypotheses, stable sort by cost), batched over images: every step is ONE set of device
  launches over all live nodes of all images (st_embedding_rows, st_gather_state, st_rnn_step,
  st_softmax_topk); only the k candidates per node come back to the host, where the reference's list
  bookkeeping (``Node`` parents, stable ``sorted``) is reproduced literally, including its float32 cost
  accumulation under NumPy 2 and the fact that nodes ending on the last iteration are never harvested.
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


class _Node:
    __slots__ = ("parent", "value", "cum_cost", "row")

    def __init__(self, parent, value, cost, row):
        self.parent, self.value, self.row = parent, value, row
        self.cum_cost = parent.cum_cost + cost if parent else cost               # beam_search.py:24

    def sequence(self):
        out, n = [], self
        while n:
            out.insert(0, n.value)
            n = n.parent
        return out


def beam_search(rnn, cnn_feature, beam_width=4, num_hypotheses=1, max_length=50, start_id=1, end_id=2):
    """beam_search.py:45-97 for every image of the batch.  Returns, per image, a list of at most
    `num_hypotheses` (token_sequence, cum_cost) pairs; the list is empty when no beam ever emitted
    `end_id` in time (the reference returns [] then)."""
    with torch.no_grad():
        st = _Stepper(rnn)
        B = cnn_feature.shape[0]
        _, state = st.step(_feat(rnn, cnn_feature), None, want_logits=False)      # state after the image-feature step (rnn.py:41,49)
        next_fringe = [[_Node(None, start_id, 0.0, b)] for b in range(B)]         # beam_search.py:66
        hyps = [[] for _ in range(B)]
        done = [False] * B
        for _ in range(max_length):                                               # beam_search.py:69
            rows, owner = [], []
            fringes = [None] * B
            for b in range(B):
                if done[b]:
                    continue
                fr = []
                for n in next_fringe[b]:
                    (hyps[b] if n.value == end_id else fr).append(n)              # beam_search.py:72-76
                if not fr:
                    done[b] = True                                                # beam_search.py:78-79 (break)
                    continue
                fringes[b] = fr
                rows += fr
                owner += [b] * len(fr)
            if not rows:
                break
            x = st.embed([n.value for n in rows])
            logits, new_state = st.step(x, st.gather(state, [n.row for n in rows]))
            tp, ti = st.topk(logits, min(beam_width, st.V), raw=False)            # descending; argsort(p)[-k:] is ascending
            r = 0
            for b in range(B):
                fr = fringes[b]
                if fr is None:
                    continue
                cand = []
                for n in fr:                                                      # beam_search.py:87-92
                    nll = -np.log(tp[r][::-1].astype(np.float32))
                    for y, c in zip(ti[r][::-1], nll):
                        cand.append(_Node(n, int(y), c, r))
                    r += 1
                next_fringe[b] = sorted(cand, key=lambda n: n.cum_cost)[:beam_width]   # beam_search.py:94 (stable)
            state = new_state
        out = []
        for b in range(B):
            hyps[b].sort(key=lambda n: n.cum_cost)                                # beam_search.py:96
            out.append([(n.sequence(), float(n.cum_cost)) for n in hyps[b][:num_hypotheses]])
        return out
