"""Latency of the trainable chain inside the software-pipelined loop (HIP events on the main stream): from the moment the step's backbone
features are available to the end of backward, against the step period.  If the two agree the loop is bound by the chain's latency
under contention with the encoder kernels, not by throughput.   usage: python tools/chain_latency.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import optim
from showtell_amd.cnn import ResNet
from showtell_amd.rnn import RNN
from showtell_amd.head import linear_bn1d
from showtell_amd.train import Trainer, synthetic_batch

class T2(Trainer):
    evs = []
    def step(self, image, caption, caption_len, upcoming=()):
        cnn, rnn = self.cnn, self.rnn
        pooled = self._backbone(image)
        upcoming = [im for im in list(upcoming)[:self.depth] if im is not None]
        for j, im in enumerate(upcoming):
            if j >= len(self._pre):
                self._prefetch(im)
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        self._apply_pending()
        self.opt.zero_grad()
        feat = linear_bn1d(pooled, cnn.linear_secondlast_layer, cnn.last_layer, cnn.training, cnn.compute_dtype)
        e[1].record()
        loss = rnn.loss(feat, caption, caption_len)
        e[2].record()
        loss.backward()
        e[3].record()
        self.reducer.start(self.opt.flat_grad)
        self.pending = True
        T2.evs.append(e)
        return loss

dev = torch.device("cuda", 0)
E, H, L, V, B = 512, 512, 5, 10000, 128
torch.manual_seed(1)
cnn = ResNet(101, E, dtype=torch.bfloat16).to(dev).train()
rnn = RNN(E, H, V, L, dtype=torch.bfloat16).to(dev).train()
opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9)
tr = T2(cnn, rnn, opt, 1)
image, caption, lens = synthetic_batch(B, V, seed=1, device=dev)
n = 40
for k in range(n):
    tr.step(image, caption, lens, upcoming=[image] * min(tr.depth, n - 1 - k))
tr.flush(); torch.cuda.synchronize()
ev = T2.evs[10:-4]
f = lambda i, j: sum(e[i].elapsed_time(e[j]) for e in ev) / len(ev)
period = sum(a[0].elapsed_time(b[0]) for a, b in zip(ev[:-1], ev[1:])) / (len(ev) - 1)
print(f"step period {period:.3f} ms; chain: sgd + zero_grad + head {f(0, 1):.3f} ms, decoder forward + loss {f(1, 2):.3f} ms, backward {f(2, 3):.3f} ms, total {f(0, 3):.3f} ms")
idle = sum(a[3].elapsed_time(b[0]) for a, b in zip(ev[:-1], ev[1:])) / (len(ev) - 1)
print(f"main stream between the end of backward and the next step's chain start (waiting for the next forward): {idle:.3f} ms")
