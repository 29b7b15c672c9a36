"""Host-side cost of one pipelined train step (bench.py's loop): wall time to ENQUEUE the steps (no synchronise inside) against
wall time including the final synchronise.  If the two agree the loop is bound by the host (Python + launch overhead), not the GPU.
usage: python tools/host_time.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import optim
from showtell_amd.cnn import ResNet
from showtell_amd.rnn import RNN
from showtell_amd.train import Trainer, synthetic_batch

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda", 0)
E, H, L, V, B = 512, 512, 5, 10000, int(os.environ.get("HB", "128"))
torch.manual_seed(1)
cnn = ResNet(101, E, dtype=torch.bfloat16).to(dev).train()
rnn = RNN(E, H, V, L, dtype=torch.bfloat16).to(dev).train()
opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9)
trainer = Trainer(cnn, rnn, opt, 1)
image, caption, lens = synthetic_batch(B, V, seed=1, device=dev)
ahead = lambda k, n: dict(upcoming=[image] * min(trainer.depth, n - 1 - k))
for k in range(10):
    trainer.step(image, caption, lens, **ahead(k, 10))
trainer.flush(); torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(steps):
    trainer.step(image, caption, lens, **ahead(k, steps))
t1 = time.perf_counter()
trainer.flush(); torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0) / steps:.3f} ms/step   with final synchronise {1e3 * (t2 - t0) / steps:.3f} ms/step   (tail after the last enqueue {1e3 * (t2 - t1):.2f} ms)")
# where the host time goes: the same loop under cProfile
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for k in range(steps):
    trainer.step(image, caption, lens, **ahead(k, steps))
pr.disable(); trainer.flush(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(18)
