"""Where the pipelined step's time goes on the MAIN stream: per step, (a) waiting for the prefetched backbone features,
(b) the trainable part (head, decoder forward / backward, optimizer) while three backbone forwards share the chip (debug aid)."""
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
image, caption, lens = synthetic_batch(B, V, seed=1, device=dev)
trainer = Trainer(cnn, rnn, opt, 1)
orig = trainer._backbone
marks = []


def probed(img):
    e0 = torch.cuda.Event(enable_timing=True); e0.record()       # main stream reaches the step
    out = orig(img)
    e1 = torch.cuda.Event(enable_timing=True); e1.record()       # ... has the features
    marks.append((e0, e1))
    return out


trainer._backbone = probed
n = 40
for k in range(10):
    trainer.step(image, caption, lens, upcoming=[image] * min(trainer.depth, 9 - k))
trainer.flush(); torch.cuda.synchronize()
marks.clear()
t0 = time.perf_counter()
for k in range(n):
    trainer.step(image, caption, lens, upcoming=[image] * min(trainer.depth, n - 1 - k))
trainer.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n * 1e3
wait = [a.elapsed_time(b) for a, b in marks[5:-5]]
work = [marks[i][1].elapsed_time(marks[i + 1][0]) for i in range(5, len(marks) - 6)]
print(f"{dt:.3f} ms/step; main stream per step: waits {sum(wait) / len(wait):.3f} ms for the features, then runs the trainable part in "
      f"{sum(work) / len(work):.3f} ms (1.27 ms alone)")
