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
        self._stashed = None
        self.param_groups = [{"params": params, "lr": lr}]
        if params[0].is_cuda:
            self._ensure_flat()

    def _ensure_flat(self):
        """(Re)build the flat buffers.  Lazy because the reference constructs its optimizer before
        moving the models to the GPU (main.py:96-111); `.cuda()` swaps every param.data."""
        params = self.params
        if self.flat is not None and all(p.data_ptr() == self.flat.data_ptr() + 4 * o for p, o in zip(params, self.offsets)):
            # the data views are intact; `module.zero_grad()` (set_to_none=True by default) or a stray `p.grad = ...` may have
            # replaced the gradient views -- re-point them, keeping whatever had been accumulated
            for p, o in zip(params, self.offsets):
                if p.grad is None or p.grad.data_ptr() != self.flat_grad.data_ptr() + 4 * o:
                    old_grad = p.grad
                    view = self.flat_grad[o:o + p.numel()].view(p.shape)
                    if old_grad is not None:
                        view.copy_(old_grad.to(view.device))
                    else:
                        view.zero_()               # set_to_none: the slice still holds the previous step's gradient
                    p.grad = view
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
        if self._stashed is not None:              # load_state_dict() came before the parameters reached the GPU (main.py:96-121)
            sd, self._stashed = self._stashed, None
            self._apply_state(sd)

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
        """Gradients stay views of the flat buffer (the all-reduce target); `set_to_none` is accepted and ignored."""
        self._ensure_flat()
        self.flat_grad.zero_()

    # ---- state_dict in torch.optim's layout (main.py:121 / utils.py:131-138): per-parameter entries keyed by index ------
    def state_dict(self):
        """{'state': {i: {<per-tensor buffers>}}, 'param_groups': [{..., 'params': [0..n-1]}]} exactly as
        torch.optim.SGD / Adam write it, so a checkpoint written here loads into the reference's optimizer and back."""
        owner = getattr(self, "_pending_owner", None)
        if owner is not None:                      # a pipelined Trainer defers optimizer.step() by one step: apply it first
            owner.flush()
        state = {}
        if self.flat is not None and self.steps > 0:
            flat_state = self._state()
            for i, (p, o) in enumerate(zip(self.params, self.offsets)):
                ent = {k: v[o:o + p.numel()].view(p.shape).clone() for k, v in flat_state.items()}
                ent.update(self._per_param_extra())
                if ent:
                    state[i] = ent
        group = {"lr": self.param_groups[0]["lr"], **self._hyper(), "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        if not isinstance(sd, dict) or "state" not in sd:
            raise ValueError("optimizer state_dict needs a 'state' entry (torch.optim layout)")
        if self.flat is None and not self.params[0].is_cuda:
            self._stashed = sd                     # applied by _ensure_flat once `.cuda()` has run (reference order, main.py:96-121)
            self._restore_hyper(sd)
            return
        self._ensure_flat()
        self._apply_state(sd)

    def _restore_hyper(self, sd):
        groups = sd.get("param_groups") or []
        if groups:
            if len(groups) != 1:
                raise ValueError("showtell_amd optimizers hold one parameter group, the checkpoint has %d" % len(groups))
            g = groups[0]
            if "params" in g and len(g["params"]) != len(self.params):
                raise ValueError("optimizer checkpoint covers %d tensors, this optimizer %d" % (len(g["params"]), len(self.params)))
            if "lr" in g:
                self.lr = self.param_groups[0]["lr"] = float(g["lr"])
            self._load_hyper(g)

    def _apply_state(self, sd):
        self._restore_hyper(sd)
        state = sd["state"]
        flat_state = self._state()
        if state and all(isinstance(k, str) and not k.isdigit() for k in state):
            # round-1 flat layout: whole-buffer tensors keyed by NAME ('momentum_buffer', 'exp_avg', ..); a torch.optim state whose
            # integer keys were stringified ('0', '1', ..: JSON round trips) takes the per-parameter path below
            for k, v in state.items():
                if k not in flat_state or flat_state[k].numel() != v.numel():
                    raise ValueError("flat optimizer state %r does not match this optimizer" % (k,))
                flat_state[k].copy_(v.to(flat_state[k].device))
            self.steps = int(sd.get("steps", 0))
            return
        steps = 0
        for i, (p, o) in enumerate(zip(self.params, self.offsets)):
            ent = state.get(i, state.get(str(i)))
            if ent is None:
                continue                            # torch.optim creates state lazily: a tensor without a gradient so far has none
            for k, buf in flat_state.items():
                if k not in ent or ent[k] is None:
                    raise ValueError("optimizer checkpoint entry %d lacks %r" % (i, k))
                t = ent[k]
                if tuple(t.shape) != tuple(p.shape):
                    raise ValueError("optimizer checkpoint entry %d: %r has shape %s, parameter %s" % (i, k, tuple(t.shape), tuple(p.shape)))
                buf[o:o + p.numel()].copy_(t.reshape(-1).to(buf.device, torch.float32))
            st = ent.get("step", None)
            steps = max(steps, int(st.item() if torch.is_tensor(st) else st) if st is not None else 1)
        self.steps = int(sd.get("steps", steps))


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
        return {"momentum": self.momentum, "dampening": 0, "weight_decay": 0, "nesterov": False}

    def _load_hyper(self, g):
        if g.get("dampening", 0) or g.get("weight_decay", 0) or g.get("nesterov", False):
            raise ValueError("showtell_amd SGD implements the reference's SGD(lr, momentum) only (main.py:98): no dampening / weight decay / nesterov")
        self.momentum = float(g.get("momentum", self.momentum))
        if self.momentum != 0 and self.buf is None and self.flat is not None:
            self.buf = torch.zeros(self.n, device=self.flat.device)

    def _per_param_extra(self):
        return {}

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
        return {"betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False}

    def _load_hyper(self, g):
        if g.get("weight_decay", 0) or g.get("amsgrad", False):
            raise ValueError("showtell_amd Adam implements the reference's Adam(lr) only (main.py:100): no weight decay / amsgrad")
        self.betas = tuple(float(b) for b in g.get("betas", self.betas))
        self.eps = float(g.get("eps", self.eps))

    def _per_param_extra(self):
        return {"step": int(self.steps)}

    def step(self):
        self._ensure_flat()
        self.steps += 1
        lr = self.param_groups[0]["lr"]
        check(lib().st_adam_step(_cp(self.flat), _cp(self.flat_grad), _cp(self.m), _cp(self.v), _cp(self.shadow), self.n, float(lr),
                                 float(self.betas[0]), float(self.betas[1]), float(self.eps), int(self.steps),
                                 float(self.grad_scale), _stream()), "st_adam_step")
        self._sync_shadow()
