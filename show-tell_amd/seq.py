"""Packed-sequence bookkeeping shared by the decoders (host side).

The reference hands ``caption_size`` to its modules as a Python list sorted in
descending order (utils.py:66-69, rnn.py:31).  From it we derive, once per distinct
length tuple, the time-major row map that ``pack_padded_sequence`` implies
(row(t, b) = off[t] + b) and keep it on the device for the kernels.
"""
import ctypes as C

import numpy as np
import torch

from ._lib import PackedSeq, ShowTellHipError

_CACHE = {}


class SeqPlan:
    def __init__(self, lens, device):
        lens = [int(l) for l in lens]
        if len(lens) == 0 or min(lens) <= 0:
            raise ShowTellHipError("caption_size must hold positive lengths")
        if any(lens[i] < lens[i + 1] for i in range(len(lens) - 1)):
            # same condition torch's pack_padded_sequence enforces (enforce_sorted=True, rnn.py:31)
            raise RuntimeError("`lengths` array must be sorted in decreasing order when `enforce_sorted` is True.")
        self.lens = lens
        self.B, self.T = len(lens), lens[0]
        bs = np.array([sum(1 for l in lens if l > t) for t in range(self.T)], dtype=np.int32)
        off = np.concatenate([[0], np.cumsum(bs)]).astype(np.int64)
        self.batch_sizes, self.off, self.ntok = bs, off, int(off[-1])
        rows_b = np.concatenate([np.arange(b, dtype=np.int32) for b in bs])
        rows_t = np.concatenate([np.full(b, t, dtype=np.int32) for t, b in enumerate(bs)])
        prev = np.where(rows_t > 0, off[np.maximum(rows_t - 1, 0)] + rows_b, 0).astype(np.int32)
        self.rows_b = torch.from_numpy(rows_b).to(device)
        self.rows_t = torch.from_numpy(rows_t).to(device)
        self.prev_row = torch.from_numpy(prev).to(device)
        self._bs_c = (C.c_int * self.T)(*[int(v) for v in bs])

    def c_struct(self, caption):
        if not caption.is_cuda:
            raise ShowTellHipError("image_caption must be on the HIP device (no CPU fallback in the MI355X build)")
        if caption.dtype != torch.int64 or not caption.is_contiguous():
            raise ShowTellHipError("image_caption must be a contiguous LongTensor (utils.py:70)")
        if caption.shape[0] != self.B or caption.shape[1] < self.T:
            raise ShowTellHipError(f"caption shape {tuple(caption.shape)} does not match caption_size (B={self.B}, T={self.T})")
        return PackedSeq(self.B, self.T, self.ntok, caption.shape[1], self._bs_c,
                         C.c_void_p(self.rows_b.data_ptr()), C.c_void_p(self.rows_t.data_ptr()),
                         C.c_void_p(self.prev_row.data_ptr()), C.c_void_p(caption.data_ptr()))


def plan_for(lens, device):
    key = (tuple(int(l) for l in lens), str(device))
    p = _CACHE.get(key)
    if p is None:
        if len(_CACHE) > 256:
            _CACHE.clear()
        p = _CACHE[key] = SeqPlan(lens, device)
    return p
