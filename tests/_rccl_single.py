"""Helper of test_gpu_rccl.py (own process: it owns a process group).  Drives Trainer through the RCCL branch of
GradAllReducer on the one GPU a test box has: the group has ONE rank (RCCL refuses two ranks on one device), but
the reducer is told world=2, so every step issues the bucketed asynchronous all-reduce on RCCL's stream beside the
pipelined backbone forwards and then scales the gradient by 1/2.  all-reduce(sum) over one rank is the identity,
so the run must equal a plain run whose learning rate is halved (SGD with momentum is linear in the gradient)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist


def run(rccl):
    from showtell_amd import optim
    from showtell_amd.cnn import ResNet
    from showtell_amd.rnn import RNN
    from showtell_amd.train import Trainer, synthetic_batch
    E = H = 256
    L, V, B = 2, 1000, 32
    torch.manual_seed(11)
    cnn = ResNet(50, E, dtype=torch.bfloat16).cuda().train()
    rnn = RNN(E, H, V, L, dtype=torch.bfloat16).cuda().train()
    opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.08 if rccl else 0.04, momentum=0.9)
    tr = Trainer(cnn, rnn, opt)
    if rccl:
        tr.reducer.world = 2
        tr.reducer.bucket = 1 << 18          # several buckets in flight
    batches = [synthetic_batch(B, V, seed=40 + i, image_size=128) for i in range(5)]
    out = []
    for i, (img, cap, lens) in enumerate(batches):
        out.append(float(tr.step(img, cap, lens, upcoming=[bb[0] for bb in batches[i + 1:i + 4]]).detach()))
        if rccl:
            assert tr.reducer.pending, "the all-reduce branch did not run"
    tr.flush()
    torch.cuda.synchronize()
    return out, [p.detach().float().cpu() for p in Trainer.trainable_params(cnn, rnn)]


if __name__ == "__main__":
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[1], RANK="0", WORLD_SIZE="1")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", rank=0, world_size=1)
    la, pa = run(True)
    lb, pb = run(False)
    dist.barrier()
    dist.destroy_process_group()
    assert np.allclose(la, lb, rtol=2e-3, atol=2e-3), (la, lb)
    for x, y in zip(pa, pb):
        assert torch.allclose(x, y, rtol=2e-2, atol=2e-3), float((x - y).abs().max())
    print("RCCL_SINGLE_OK", la, lb)
