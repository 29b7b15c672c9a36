"""bench.py's pipelined loop with the trainable part on a high-priority stream (or not): ms per step.  usage: python tools/time_step_prio.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import optim
from showtell_amd.cnn import ResNet
from showtell_amd.rnn import RNN
from showtell_amd.train import Trainer, synthetic_batch
dev = torch.device("cuda", 0)
E, H, L, V, B = 512, 512, 5, 10000, 128
print("priority range", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "?")
for prio in (None, -1, None, -1):
    torch.manual_seed(1)
    cnn = ResNet(101, E, dtype=torch.bfloat16).to(dev).train()
    rnn = RNN(E, H, V, L, dtype=torch.bfloat16).to(dev).train()
    opt = optim.SGD(Trainer.trainable_params(cnn, rnn), lr=0.01, momentum=0.9)
    trainer = Trainer(cnn, rnn, opt, 1)
    image, caption, lens = synthetic_batch(B, V, seed=1, device=dev)
    ahead = lambda k, n: dict(upcoming=[image] * min(trainer.depth, n - 1 - k))
    st = torch.cuda.Stream(priority=prio) if prio is not None else torch.cuda.current_stream()
    st.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(st):
        for k in range(10):
            trainer.step(image, caption, lens, **ahead(k, 10))
        trainer.flush(); torch.cuda.synchronize()
        n = 50
        t0 = time.perf_counter()
        for k in range(n):
            trainer.step(image, caption, lens, **ahead(k, n))
        trainer.flush(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"main stream priority {prio}: {1e3 * dt / n:.3f} ms/step  {B * n / dt:.0f} img/s", flush=True)
    del trainer, cnn, rnn, opt
