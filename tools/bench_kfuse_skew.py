"""Does the RELATIVE placement of the three 196-MiB streams of the fused block-end loader matter?  raw / identity / x_out carved out of
one allocation at 196 MiB + skew apart (the engine's workspace places them exactly 196 MiB apart at B = 128)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops

B, H, C, N = 128, 56, 256, 64
nel = B * H * H * C
ntw = ops.conv1x1_kfuse_supported(C, N)
w = torch.randn(N, C, 1, 1, device="cuda") / C ** 0.5
wf = ops.pack_conv_weight_frag(w, ntw)
gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
st = torch.zeros(4, 2 * C, device="cuda"); st[0, C:] = float(B * H * H)
n = float(B * H * H)
so = torch.zeros(4, 2 * N, device="cuda")
y = torch.empty(B, H, H, N, device="cuda", dtype=torch.bfloat16)
bn = dict(stats=st, gamma=gam, beta=bet, count=n, replicas=4)
for skew in [0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096 * 3 + 256]:
    flat = torch.empty(3 * (nel + skew // 2) + 1024, device="cuda", dtype=torch.bfloat16).normal_()
    bufs = [flat[i * (nel + skew // 2): i * (nel + skew // 2) + nel].view(B, H, H, C) for i in range(3)]
    raw, ident, x = bufs

    def fused():
        ops.conv1x1_kfuse(raw, ident, wf, bn, N=N, stats=so, stats_replicas=4, x_out=x, out=y)

    def sep():
        ops.bn_act(raw, gam, bet, stats=st, count=n, relu=True, res=ident, out=raw, stats_replicas=4)
        ops.conv1x1_wreg(raw, wf, N, stats=so, stats_replicas=4, out=y)
    res = []
    for f in (sep, fused):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"skew {skew:8d} B: in-place bn_act + wreg {res[0]:.1f} us, fused {res[1]:.1f} us", flush=True)
    del flat, bufs, raw, ident, x
