"""Helper of test_gpu_dp.py: ONE rank of a 2-rank data-parallel training run (own process, gloo process group, both ranks on the
one GPU of the test box).  Each rank builds the model from its OWN seed (rank 0's weights must win through Trainer's
broadcast), trains `steps` Trainer steps on its OWN minibatches and rank 0 saves the final trainable parameters."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

E = H = 64
L, V, B, SIZE, STEPS = 2, 120, 4, 96, 3


def build(seed, dtype=torch.float32):
    from showtell_amd import optim
    from showtell_amd.cnn import ResNet
    from showtell_amd.rnn import RNN
    from showtell_amd.train import Trainer
    torch.manual_seed(seed)
    cnn = ResNet(18, E, dtype=dtype).cuda().train()
    rnn = RNN(E, H, V, L, dtype=dtype).cuda().train()
    opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.05, momentum=0.9, shadow_dtype=None if dtype == torch.float32 else dtype)
    return cnn, rnn, opt


def batches(rank):
    from showtell_amd.train import synthetic_batch
    return [synthetic_batch(B, V, seed=300 + 10 * rank + i, image_size=SIZE, mean=6, std=1.5, lo=4, hi=9) for i in range(STEPS)]


if __name__ == "__main__":
    port, rank, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, RANK=str(rank), WORLD_SIZE="2", LOCAL_RANK="0")
    torch.cuda.set_device(0)
    dist.init_process_group(backend="gloo", rank=rank, world_size=2)
    from showtell_amd.train import Trainer
    cnn, rnn, opt = build(seed=50 + rank)              # different initial weights per rank: the broadcast must equalise them
    probe = torch.randn(2, 3, SIZE, SIZE, generator=torch.Generator().manual_seed(9)).cuda()
    cnn.eval()
    with torch.no_grad():
        cnn.backbone_features(probe)                   # a forward BEFORE Trainer(): the backbone packs THIS rank's own filters (its cache
    cnn.train()                                        # is keyed on tensor versions; the broadcast below must invalidate it)
    tr = Trainer(cnn, rnn, opt, world_size=2)
    losses = []
    for img, cap, lens in batches(rank):
        losses.append(float(tr.step(img, cap, lens).detach()))
        assert tr.reducer.pending, "the all-reduce branch did not run"
    tr.flush()
    torch.cuda.synchronize()
    flat = opt.flat.detach().cpu().numpy()
    both = [None, None]
    dist.all_gather_object(both, flat)
    assert np.array_equal(both[0], both[1]), "ranks ended with different parameters"
    with torch.no_grad():
        pf = cnn.backbone_features(probe).detach().cpu().numpy()   # train-mode BN: a function of the frozen filters / gamma / beta only
    feats = [None, None]
    dist.all_gather_object(feats, pf)
    # (float atomics sum the batch statistics in a different order per process: equal to fp32 rounding, not bit for bit)
    assert np.allclose(feats[0], feats[1], rtol=1e-3, atol=1e-4 * float(np.abs(feats[0]).max())), "the frozen backbones of the two ranks differ: a rank kept its pre-broadcast packed filters"
    if rank == 0:
        np.savez(out, flat=flat, losses=np.array(losses), rm=cnn.last_layer.running_mean.detach().cpu().numpy())
    dist.barrier()
    dist.destroy_process_group()
    print("DP_RANK_OK", rank, losses)
