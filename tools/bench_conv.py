"""Micro-benchmark of st_conv on the ResNet-101 conv shapes (B=128, bf16): per-variant time, TFLOP/s, GB/s."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
from showtell_amd._lib import lib

B = 128
SHAPES = [  # cin, cout, k, s, hin
    (256, 256, 3, 1, 14), (256, 1024, 1, 1, 14), (1024, 256, 1, 1, 14), (64, 256, 1, 1, 56),
    (128, 512, 1, 1, 28), (64, 64, 3, 1, 56), (128, 128, 3, 1, 28), (512, 512, 3, 1, 7), (512, 2048, 1, 1, 7),
]
VARIANTS = [("kc8 4w", 0, 8, 0), ("kc8 8w", 0, 8, 1), ("kc4 8w", 0, 4, 1), ("256x128", 0, 8, 2)]
if len(sys.argv) > 1:
    VARIANTS = [v for v in VARIANTS if any(v[0].startswith(a) for a in sys.argv[1:])]
dt = torch.bfloat16
for cin, cout, k, s, h in SHAPES:
    x = torch.randn(B, h, h, cin, device="cuda").to(dt)
    w = (torch.randn(cout, k * k * cin, device="cuda") / (k * k * cin) ** 0.5).to(dt)
    KO = int(os.environ.get("KORDER", "0")) if k > 1 and cin % 64 == 0 else 0
    stats = torch.zeros(2 * cout, device="cuda")
    ho = (h + 2 * (k // 2) - k) // s + 1
    flops = 2.0 * B * ho * ho * cout * cin * k * k
    byts = x.numel() * 2 + w.numel() * 2 + B * ho * ho * cout * 2
    line = f"{cin:5d}->{cout:5d} k{k} s{s} @{h:3d}: "
    for name, ring, kc, w8 in VARIANTS:
        lib().st_tune(ring, kc, w8)
        out = ops.conv_nhwc(x, w, k, k, s, k // 2, stats=stats, k_order=KO)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ops.conv_nhwc(x, w, k, k, s, k // 2, stats=stats, out=out, k_order=KO)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        line += f"| {name}: {us:6.1f}us {flops/us/1e6:5.0f}TF {byts/us/1e3:5.0f}GB/s "
    print(line, flush=True)
