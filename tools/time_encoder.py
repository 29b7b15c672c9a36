"""Time the ResNet-101 backbone forward at B=128 (debug aid; bench.py is the contract)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.cnn import ResNet
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for dtype in (torch.bfloat16,):
    for train in (True, False):
        m = ResNet(101, 512, dtype=dtype).cuda().train(train)
        x = torch.randn(B, 3, 224, 224, device="cuda")
        for _ in range(3):
            m.backbone_features(x)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            m.backbone_features(x)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        print(f"dtype={dtype} train={train} B={B}: {ms:.2f} ms/fwd  {B/ms*1e3:.0f} img/s  {15.6e9*B/ms/1e9:.1f} TFLOP/s")
