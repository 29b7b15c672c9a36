"""Back-to-back timing of st_conv1x1_kstream against st_conv on the long-K pointwise geometries of ResNet-101 at B=128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
B = 128
for (h, c, n, s, cnt) in ((14, 1024, 256, 1, 22), (14, 1024, 512, 1, 1), (14, 1024, 2048, 2, 1), (7, 2048, 512, 1, 2)):
    x = torch.randn(B, h, h, c, device="cuda").bfloat16()
    w = torch.randn(n, c, 1, 1, device="cuda") / c ** 0.5
    wf = ops.pack_conv_weight_frag(w, 4)
    wg = ops.pack_conv_weight(w, torch.bfloat16)
    R = 4
    st = torch.zeros(R, 2 * n, device="cuda")
    ho = (h - 1) // s + 1
    y = torch.empty(B, ho, ho, n, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * B * ho * ho * c * n
    def t(fn, it=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    t_k = t(lambda: ops.conv1x1_kstream(x, wf, n, stride=s, stats=st, stats_replicas=R, out=y))
    t_g = t(lambda: ops.conv_nhwc(x, wg, 1, 1, s, 0, stats=st, stats_replicas=R, out=y))
    print(f"1x1 {c:4d}->{n:4d} s{s} @{h:2d} x{cnt:2d}: kstream {t_k:6.1f} us ({fl / t_k / 1e6:4.0f} TF)  |  igemm {t_g:6.1f} us ({fl / t_g / 1e6:4.0f} TF)", flush=True)
