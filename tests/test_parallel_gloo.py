"""N>1 data-parallel plumbing on CPU: world_size-2 gloo processes drive the same GradAllReducer
that bench.py uses with RCCL.  (The HIP kernels themselves need a GPU; what is covered here is the
bucketing, the asynchronous start/finish protocol and the mean-over-ranks semantics.)"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, bucket, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from showtell_amd import parallel
    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    flat = torch.randn(n, generator=g)
    red = parallel.GradAllReducer(bucket_elems=bucket)
    red.start(flat)                       # async: independent work may run here
    independent = torch.ones(10).sum()    # stands in for the next step's frozen backbone forward
    scale = red.finish()
    assert independent.item() == 10 and not red.pending
    np.save(os.path.join(out_dir, f"r{rank}.npy"), (flat * scale).numpy())
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,bucket", [(1000, 256), (77, 1 << 20), (4096, 1024)])
def test_allreduce_mean_two_ranks(tmp_path, n, bucket):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, n, bucket, str(tmp_path)), nprocs=world, join=True)
    got = [np.load(tmp_path / f"r{r}.npy") for r in range(world)]
    ref = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world)).numpy() / world
    for g_ in got:                         # every rank ends with the same averaged gradient
        np.testing.assert_allclose(g_, ref, rtol=1e-6, atol=1e-6)


def test_single_process_is_a_noop():
    from showtell_amd.parallel import GradAllReducer
    red = GradAllReducer(world_size=1)
    t = torch.arange(5.0)
    red.start(t)
    assert red.finish() == 1.0 and torch.equal(t, torch.arange(5.0))


def _bcast_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from showtell_amd import parallel
    parallel.init_from_env("gloo")
    torch.manual_seed(rank)                                    # every rank starts from DIFFERENT weights and buffers
    m = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3))
    with torch.no_grad():
        m[1].running_mean.normal_(); m[1].num_batches_tracked.fill_(7 + rank)
    parallel.broadcast_state([m], None)
    torch.save(m.state_dict(), os.path.join(out_dir, f"sd{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_state_equalises_replicas(tmp_path):
    """Trainer(world>1) starts every replica from rank 0's parameters AND buffers (what DDP does at construction)."""
    world, port = 2, _free_port()
    mp.spawn(_bcast_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    a, b = (torch.load(tmp_path / f"sd{r}.pt", weights_only=True) for r in range(world))
    torch.manual_seed(0)
    ref = torch.nn.Sequential(torch.nn.Linear(5, 3), torch.nn.BatchNorm1d(3)).state_dict()
    for k in a:
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(a["0.weight"], ref["0.weight"]) and int(a["1.num_batches_tracked"]) == 7
