"""Time st_conv_c3c1 alone at the engine's shapes (B = 128): the 14 x 14 geometry (256 -> 1024 -> 256) and the 28 x 28 one
(128 -> 512 -> 128), train and eval forms.  Debug aid; ST_C3C1_L2 picks the 28 x 28 form.  usage: python tools/time_c3c1.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for C1, C2, H in ((256, 1024, 14), (128, 512, 28)):
    g = torch.Generator().manual_seed(0)
    x2 = torch.randn(B, H, H, C1, generator=g).bfloat16().cuda()
    ident = torch.relu(torch.randn(B, H, H, C2, generator=g)).bfloat16().cuda()
    w3 = ops.pack_conv_weight_frag((torch.randn(C2, C1, 1, 1, generator=g) / C1 ** 0.5).bfloat16().cuda(), 2)
    w1 = ops.pack_conv_weight_frag((torch.randn(C1, C2, 1, 1, generator=g) / C2 ** 0.5).bfloat16().cuda(), C1 // 64)
    n = float(B * H * H)
    x2f = x2.float().reshape(-1, C1)
    st2 = torch.zeros(4, 2 * C1, device="cuda"); st2[0] = torch.cat([x2f.sum(0), (x2f * x2f).sum(0)])
    s3 = torch.zeros(4, 2 * C2, device="cuda"); s3[0, C2:] = n
    one = lambda c: torch.ones(c, device="cuda")
    zero = lambda c: torch.zeros(c, device="cuda")
    s1 = torch.zeros(4, 2 * C1, device="cuda")
    xo, yo = torch.empty_like(ident), torch.empty_like(x2)
    forms = {
        "train": lambda: ops.conv_c3c1(x2, w3, ident, w1, bn2=dict(stats=st2, gamma=one(C1), beta=zero(C1), count=n, replicas=4),
                                       bn3=dict(stats=s3, gamma=one(C2), beta=zero(C2), replicas=4), count=n, stats=s1, stats_replicas=4, x_out=xo, out=yo),
        "eval": lambda: ops.conv_c3c1(x2, w3, ident, w1, scale3=one(C2), shift3=zero(C2), scale1=one(C1), shift1=zero(C1), x_out=xo, out=yo),
    }
    for name, f in forms.items():
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        it = 50
        e0.record()
        for _ in range(it):
            f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / it * 1e3
        byts = n * (C1 + 2 * C2 + C1) * 2
        print(f"c3c1 {C1}->{C2}->{C1} @{H}x{H} B={B} {name}: {us:.1f} us/launch (back to back, incl. host wrapper)  {byts / us / 1e6:.2f} TB/s algorithmic", flush=True)
