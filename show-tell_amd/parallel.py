"""Data parallelism: one process per GPU, gradients averaged with RCCL all-reduce over xGMI.

The reference has no multi-GPU code; this is the DP path BASELINE config 4 asks
for.  Design for the MI355X node (xGMI is point-to-point, 7 links/GPU, no switch):
  * all trainable gradients already live in ONE flat fp32 buffer (optim._FlatOptimizer), so the
    exchange is a handful of large messages, not ~28 small ones;
  * buckets are issued asynchronously on RCCL's stream right after backward and are only waited
    for when the NEXT step needs the updated parameters -- i.e. after the next step's frozen
    backbone forward (94 % of the step), which does not depend on them.  The all-reduce
    (76.7 MB fp32 for the GRU model) is therefore hidden behind convolution work by construction;
  * semantics: mean over ranks of the per-rank mean loss gradients (torch DDP semantics).  BatchNorm
    statistics stay local to each rank (DDP default); running buffers are rank-0's at checkpoint.
Works with the `nccl` backend (= RCCL on ROCm) on GPUs and with `gloo` on CPU tensors (tests).
"""
import os

import torch
import torch.distributed as dist

BUCKET_ELEMS = 8 * 1024 * 1024   # 32 MB fp32 per message: large enough to run at link bandwidth


def init_from_env(backend=None, device_index=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, 0
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
    if backend is None:
        backend = "nccl" if torch.cuda.is_available() else "gloo"
    if torch.cuda.is_available():
        torch.cuda.set_device(local if device_index is None else device_index)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def broadcast_state(modules, optimizer=None, src=0):
    """Rank `src`'s parameters and buffers to every rank (what DDP does at construction): replicas must start equal even
    when the caller did not seed every rank identically.  In place, so views into the optimizer's flat buffer stay views."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    if optimizer is not None:
        optimizer._ensure_flat()
    with torch.no_grad():
        for m in modules:
            for t in list(m.parameters()) + list(m.buffers()):
                if t.numel() == 0:
                    continue
                if t.dim() == 0:                      # 0-d buffers (num_batches_tracked) may be views of a counter array
                    v = t.data.reshape(1).clone()
                    dist.broadcast(v, src)
                    t.data.copy_(v.reshape(()))
                else:
                    dist.broadcast(t.data, src)
                # the write went through `.data`: bump the version counter so that caches keyed on (version, data_ptr) -- the
                # backbone's packed bf16 / fragment-major filter copies (cnn.py:pack_weights), the optimizer's bf16 shadows --
                # see that the tensor changed even when a forward had already run on this rank's own initial weights
                torch.autograd.graph.increment_version(t)
            bb = getattr(m, "_bb", None)              # frozen backbone: drop the packed copies outright (re-packed on the next forward)
            if bb is not None:
                bb.packed = None
                bb.packed_key = None
    if optimizer is not None:
        optimizer._sync_shadow(initial=True)         # the bf16 copies the kernels read


class GradAllReducer:
    """Asynchronous bucketed all-reduce(mean) of a flat gradient buffer."""

    def __init__(self, world_size=None, bucket_elems=BUCKET_ELEMS):
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucket = bucket_elems
        self.pending = []

    def start(self, flat_grad):
        """Issue the all-reduce; returns immediately (GPU) so that independent work overlaps it."""
        if self.world == 1:
            return
        n = flat_grad.numel()
        for o in range(0, n, self.bucket):
            chunk = flat_grad[o:min(n, o + self.bucket)]
            self.pending.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, async_op=True))

    def finish(self):
        """Wait for the exchange; the caller then scales by 1/world (folded into the optimizer kernel)."""
        for w in self.pending:
            w.wait()
        self.pending = []
        return 1.0 / self.world
