"""``torch.optim.SGD(params, lr, momentum)`` / ``torch.optim.Adam(params, lr)`` as used by the
reference (main.py:96-100,152), re-done as ONE multi-tensor HIP launch per step.

All parameters handed to the optimizer are moved into one flat fp32 buffer (each
``param.data`` / ``param.grad`` becomes a view), so a step is a single HBM-bound pass that
also refreshes the bf16 shadow copies the MFMA kernels read, and a data-parallel job can
all-reduce the flat gradient in a few large messages (see parallel.py).
"""
import ctypes as C
import math

import torch

from ._lib import check, lib
from .rnn import _cp, _stream


class _FlatOptimizer:
    def __init__(self, params, lr, shadow_dtype=torch.bfloat16):
        params = [p for p in params]
        if len(params) == 0:
            raise ValueError("optimizer got an empty parameter list")
        self.params, self.lr, self.shadow_dtype = params, lr, shadow_dtype
        self.flat = None
        self.steps = 0
        self.grad_scale = 1.0
        self.param_groups = [{"params": params, "lr": lr}]
        if params[0].is_cuda:
            self._ensure_flat()

    def _ensure_flat(self):
        """(Re)build the flat buffers.  Lazy because the reference constructs its optimizer before
        moving the models to the GPU (main.py:96-111); `.cuda()` swaps every param.data."""
        params = self.params
        if self.flat is not None and all(p.data_ptr() == self.flat.data_ptr() + 4 * o for p, o in zip(params, self.offsets)):
            return
        dev = params[0].device
        if dev.type != "cuda":
            raise ValueError("showtell_amd optimizers need parameters on a HIP device (no CPU path in the MI355X build)")
        offs, n = [], 0
        for p in params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4          # keep every tensor 16-byte aligned inside the flat buffer
        old_state = self._state() if self.flat is not None else {}
        self.n, self.offsets = n, offs
        self.flat = torch.zeros(n, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(n, device=dev, dtype=torch.float32)
        self.shadow = torch.zeros(n, device=dev, dtype=self.shadow_dtype) if self.shadow_dtype is not None else None
        for p, o in zip(params, offs):
            k = p.numel()
            self.flat[o:o + k].copy_(p.data.reshape(-1))
            old_grad = p.grad
            p.data = self.flat[o:o + k].view(p.shape)
            p.grad = self.flat_grad[o:o + k].view(p.shape)
            if old_grad is not None:
                p.grad.copy_(old_grad.to(dev))
        self._alloc_state(dev, old_state)
        self._sync_shadow(initial=True)

    def _sync_shadow(self, initial=False):
        if self.shadow is None:
            return
        if initial:
            from . import ops
            ops.cast(self.flat, self.shadow.dtype, out=self.shadow)
        for p, o in zip(self.params, self.offsets):
            torch.autograd.graph.increment_version(p)
            p._st_shadow = self.shadow[o:o + p.numel()].view(p.shape)
            p._st_shadow_ver, p._st_shadow_ptr = p._version, p.data_ptr()

    def zero_grad(self, set_to_none=False):
        self._ensure_flat()
        self.flat_grad.zero_()

    def state_dict(self):
        return {"state": {k: v for k, v in self._state().items()}, "param_groups": [{"lr": self.lr, **self._hyper()}],
                "steps": self.steps}

    def load_state_dict(self, sd):
        for k, v in sd["state"].items():
            self._state()[k].copy_(v)
        self.steps = int(sd.get("steps", 0))


class SGD(_FlatOptimizer):
    def __init__(self, params, lr, momentum=0.0, shadow_dtype=torch.bfloat16):
        self.momentum = momentum
        self.buf = None
        super().__init__(params, lr, shadow_dtype)

    def _alloc_state(self, dev, old):
        self.buf = torch.zeros(self.n, device=dev) if self.momentum != 0 else None
        if self.buf is not None and old.get("momentum_buffer") is not None and old["momentum_buffer"].numel() == self.n:
            self.buf.copy_(old["momentum_buffer"])

    def _state(self):
        return {"momentum_buffer": self.buf} if self.buf is not None else {}

    def _hyper(self):
        return {"momentum": self.momentum}

    def step(self):
        self._ensure_flat()
        lr = self.param_groups[0]["lr"]
        check(lib().st_sgd_step(_cp(self.flat), _cp(self.flat_grad), _cp(self.buf), _cp(self.shadow), self.n, float(lr),
                                float(self.momentum), int(self.steps == 0), float(self.grad_scale), _stream()), "st_sgd_step")
        self.steps += 1
        self._sync_shadow()


class Adam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, shadow_dtype=torch.bfloat16):
        self.betas, self.eps = betas, eps
        self.m = self.v = None
        super().__init__(params, lr, shadow_dtype)

    def _alloc_state(self, dev, old):
        self.m, self.v = torch.zeros(self.n, device=dev), torch.zeros(self.n, device=dev)
        for k, t in (("exp_avg", self.m), ("exp_avg_sq", self.v)):
            if old.get(k) is not None and old[k].numel() == self.n:
                t.copy_(old[k])

    def _state(self):
        return {"exp_avg": self.m, "exp_avg_sq": self.v} if self.m is not None else {}

    def _hyper(self):
        return {"betas": self.betas, "eps": self.eps}

    def step(self):
        self._ensure_flat()
        self.steps += 1
        lr = self.param_groups[0]["lr"]
        check(lib().st_adam_step(_cp(self.flat), _cp(self.flat_grad), _cp(self.m), _cp(self.v), _cp(self.shadow), self.n, float(lr),
                                 float(self.betas[0]), float(self.betas[1]), float(self.eps), int(self.steps),
                                 float(self.grad_scale), _stream()), "st_adam_step")
        self._sync_shadow()
