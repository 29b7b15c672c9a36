"""Back-to-back timing of st_conv1x1_wreg against st_conv on the pointwise geometries of ResNet-101 at B=128 (K <= 512)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
B = 128
for (h, c, n, s, cnt) in ((56, 64, 64, 1, 1), (56, 64, 256, 1, 4), (56, 256, 64, 1, 2), (56, 256, 128, 1, 1), (56, 256, 512, 2, 1), (28, 128, 512, 1, 4),
                          (28, 512, 128, 1, 3), (28, 512, 256, 1, 1), (28, 512, 1024, 2, 1), (14, 256, 1024, 1, 23), (7, 512, 2048, 1, 3)):
    x = torch.randn(B, h, h, c, device="cuda").bfloat16()
    w = torch.randn(n, c, 1, 1, device="cuda") / c ** 0.5
    wf = ops.pack_conv_weight_frag(w, ops.conv1x1_wreg_supported(c, n))
    wg = ops.pack_conv_weight(w, torch.bfloat16)
    R = 4
    st = torch.zeros(R, 2 * n, device="cuda")
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    x2 = x.float().reshape(-1, c)
    ist = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
    cnt_in = float(B * h * h)
    ho = (h - 1) // s + 1
    y = torch.empty(B, ho, ho, n, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * B * ho * ho * c * n
    byts = (x.numel() // (s * s) + y.numel()) * 2
    def t(fn, it=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    t_w = t(lambda: ops.conv1x1_wreg(x, wf, n, stride=s, stats=st, stats_replicas=R, out=y))
    t_wx = t(lambda: ops.conv1x1_wreg(x, wf, n, stride=s, stats=st, stats_replicas=R, out=y, in_bn=dict(stats=ist, gamma=g, beta=b, count=cnt_in)))
    t_g = t(lambda: ops.conv_nhwc(x, wg, 1, 1, s, 0, stats=st, stats_replicas=R, out=y))
    t_gx = t(lambda: ops.conv_nhwc(x, wg, 1, 1, s, 0, stats=st, stats_replicas=R, out=y, in_bn=dict(stats=ist, gamma=g, beta=b, count=cnt_in)))
    print(f"1x1 {c:4d}->{n:4d} s{s} @{h:2d} x{cnt:2d}: wreg {t_w:6.1f} us ({fl / t_w / 1e6:4.0f} TF {byts / t_w / 1e3:5.0f} GB/s)  wreg+bn {t_wx:6.1f}  |  igemm {t_g:6.1f} us  igemm+bn {t_gx:6.1f}", flush=True)
