"""Does running two independent backbone forwards on two HIP streams raise throughput (tail filling)?  Debug aid."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.cnn import ResNet
B = 128
NS = 4
ms = [ResNet(101, 512, dtype=torch.bfloat16).cuda().train() for _ in range(NS)]
xs = [torch.randn(B, 3, 224, 224, device="cuda") for _ in range(NS)]
ss = [torch.cuda.Stream() for _ in range(NS)]
for m, x in zip(ms, xs):
    for _ in range(2): m.backbone_features(x)
torch.cuda.synchronize()
def run(nstream, n=12):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n):
        k = i % nstream
        ss[k].wait_event(e0) if i < nstream else None
        with torch.cuda.stream(ss[k]):
            ms[k].backbone_features(xs[k])
    for k in range(nstream):
        torch.cuda.current_stream().wait_stream(ss[k])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for ns in (1, 2, 3, 4, 2, 3):
    print(f"{ns} stream(s): {run(ns):.2f} ms per forward", flush=True)
