"""Pipelined training step with the trainable part (main chain) on a HIGH-priority stream / the backbone side streams on LOW priority
(debug aid: is the step bound by the main chain's small kernels queueing behind backbone workgroups?)."""
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
lo, hi = torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else (0, -1)
print("priority range", lo, hi, flush=True)
for mode in (sys.argv[1:] or ["default"]):   # one mode per process: the backbone keeps at most four per-stream workspaces
    trainer = Trainer(cnn, rnn, opt, 1)
    if "side-low" in mode:
        trainer._side = [torch.cuda.Stream(priority=0) for _ in range(trainer.depth)]
    else:
        trainer._side = [torch.cuda.Stream(priority=-1) for _ in range(trainer.depth)] if mode == "all-high" else []
    main = torch.cuda.Stream(priority=-1) if "main-high" in mode else torch.cuda.current_stream()
    n = 40
    with torch.cuda.stream(main):
        for k in range(10):
            trainer.step(image, caption, lens, upcoming=[image] * min(trainer.depth, 9 - k))
        trainer.flush(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            trainer.step(image, caption, lens, upcoming=[image] * min(trainer.depth, n - 1 - k))
        trainer.flush(); torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(f"{mode}: {dt / n * 1e3:.3f} ms/step", flush=True)
