"""Back-to-back timing of st_conv3x3_img against st_conv on the four bottleneck 3x3 geometries of ResNet-101 at B=128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for (h, c) in ((14, 256), (28, 128), (56, 64), (7, 512)):
    x = torch.randn(B, h, h, c, device="cuda").bfloat16()
    w = torch.randn(c, c, 3, 3, device="cuda") / (9 * c) ** 0.5
    wf = ops.pack_conv_weight_frag(w, ops.conv3x3_img_supported(h, h, c, c))
    wg = ops.pack_conv_weight(w, torch.bfloat16, k_order=1)
    st = torch.zeros(16, 2 * c, device="cuda")
    g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
    x2 = x.float().reshape(-1, c)
    ist = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
    n = float(B * h * h)
    y = torch.empty(B, h, h, c, device="cuda", dtype=torch.bfloat16)
    fl = 2.0 * B * h * h * c * c * 9
    def t(fn, it=20):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3
    t_img = t(lambda: ops.conv3x3_img(x, wf, c, stats=st, stats_replicas=16, out=y))
    t_imgx = t(lambda: ops.conv3x3_img(x, wf, c, stats=st, stats_replicas=16, out=y, in_bn=dict(stats=ist, gamma=g, beta=b, count=n)))
    t_gem = t(lambda: ops.conv_nhwc(x, wg, 3, 3, 1, 1, stats=st, stats_replicas=16, out=y, k_order=1))
    t_bn = t(lambda: ops.bn_act(x, g, b, stats=ist, count=n, relu=True, out=y))
    print(f"3x3 {c:4d}->{c:4d} @{h:2d}: img {t_img:6.1f} us ({fl / t_img / 1e6:5.0f} TF)  img+bn {t_imgx:6.1f} us  |  igemm {t_gem:6.1f} us ({fl / t_gem / 1e6:5.0f} TF) + bn_act {t_bn:5.1f} us", flush=True)
