"""Is the pipelined training step bound by the GPU or by the host thread that issues it?  Times the host-side cost of
Trainer.step (no synchronisation inside the loop, the GPU queue is deep) beside the synchronised rate (debug aid)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import optim
from showtell_amd.cnn import ResNet
from showtell_amd.rnn import RNN
from showtell_amd.train import Trainer, synthetic_batch

dev = torch.device("cuda", 0)
E, H, L, V, B = 512, 512, 5, 10000, 128
torch.manual_seed(1)
cnn = ResNet(101, E, dtype=torch.bfloat16).to(dev).train()
rnn = RNN(E, H, V, L, dtype=torch.bfloat16).to(dev).train()
opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9)
trainer = Trainer(cnn, rnn, opt, 1)
image, caption, lens = synthetic_batch(B, V, seed=1, device=dev)
for pipe in (True, False):
    n = 40
    for k in range(10):
        trainer.step(image, caption, lens, upcoming=[image] * min(trainer.depth, 9 - k) if pipe else ())
    trainer.flush(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    host = []
    for k in range(n):
        h0 = time.perf_counter()
        trainer.step(image, caption, lens, upcoming=[image] * min(trainer.depth, n - 1 - k) if pipe else ())
        host.append(time.perf_counter() - h0)
    t_issue = time.perf_counter() - t0
    trainer.flush(); torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    host.sort()
    print(f"pipelined={pipe}: {t_all / n * 1e3:.3f} ms/step synchronised; host issue {t_issue / n * 1e3:.3f} ms/step "
          f"(median call {host[n // 2] * 1e3:.3f}, p90 {host[int(n * 0.9)] * 1e3:.3f}, min {host[0] * 1e3:.3f})", flush=True)
