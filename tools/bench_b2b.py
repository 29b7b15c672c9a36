"""One 56 x 56 block boundary at B = 128: conv3 (64 -> 256) + block-end pass + next conv1 (256 -> 64 | 128) as
(a) st_conv1x1_wreg, st_bn_act, st_conv1x1_wreg; (b) st_conv1x1_wreg, st_conv1x1_kfuse; (c) statistics-only st_conv1x1_wreg, st_conv_b2b
(debug aid; bench.py is the contract)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops

B, H, C1, C2 = 128, 56, 64, 256
for N, idbn in ((64, False), (64, True), (128, False)):
    raw2 = torch.randn(B, H, H, C1, device="cuda").bfloat16()
    ident = torch.randn(B, H, H, C2, device="cuda").bfloat16()
    w3f = ops.pack_conv_weight_frag(torch.randn(C2, C1, 1, 1, device="cuda") / 8, ops.conv1x1_wreg_supported(C1, C2))
    w1f = ops.pack_conv_weight_frag(torch.randn(N, C2, 1, 1, device="cuda") / 16, ops.conv1x1_wreg_supported(C2, N))
    g2, b2 = torch.ones(C1, device="cuda"), torch.zeros(C1, device="cuda")
    g3, b3 = torch.ones(C2, device="cuda"), torch.zeros(C2, device="cuda")
    n = float(B * H * H)
    s2 = torch.zeros(4, 2 * C1, device="cuda"); s2[0, C1:] = n
    s3 = torch.zeros(4, 2 * C2, device="cuda")
    sy = torch.zeros(4, 2 * N, device="cuda")
    raw3 = torch.empty(B, H, H, C2, device="cuda", dtype=torch.bfloat16)
    x = torch.empty_like(raw3); y = torch.empty(B, H, H, N, device="cuda", dtype=torch.bfloat16)
    bn2 = dict(stats=s2, gamma=g2, beta=b2, count=n, replicas=4)
    bn3 = dict(stats=s3, gamma=g3, beta=b3, count=n, replicas=4)
    idb = dict(stats=s3, gamma=g3, beta=b3, replicas=4) if idbn else None
    rb = dict(stats=s3, gamma=g3, beta=b3, stats_replicas=4) if idbn else None

    def a():
        s3.zero_()
        ops.conv1x1_wreg(raw2, w3f, C2, stats=s3, stats_replicas=4, in_bn=bn2, out=raw3)
        ops.bn_act(raw3, g3, b3, stats=s3, count=n, relu=True, res=ident, res_bn=rb, out=x, stats_replicas=4)
        ops.conv1x1_wreg(x, w1f, N, stats=sy, stats_replicas=4, out=y)

    def b():
        s3.zero_()
        ops.conv1x1_wreg(raw2, w3f, C2, stats=s3, stats_replicas=4, in_bn=bn2, out=raw3)
        ops.conv1x1_kfuse(raw3, ident, w1f, bn3, N=N, id_bn=idb, stats=sy, stats_replicas=4, x_out=x, out=y)

    def c():
        s3.zero_()
        ops.conv1x1_wreg(raw2, w3f, C2, stats=s3, stats_replicas=4, in_bn=bn2, stats_only=True)
        ops.conv_b2b(raw2, w3f, ident, w1f, N, dict(stats=s2, gamma=g2, beta=b2, replicas=4), dict(stats=s3, gamma=g3, beta=b3, replicas=4), n,
                     id_bn=idb, stats=sy, stats_replicas=4, x_out=x, out=y)

    res = []
    for f in (a, b, c):
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 20 * 1e3)
    print(f"56x56 64->256->{N} id_bn={idbn}: three kernels {res[0]:.1f} us | conv3 + fused loader {res[1]:.1f} us | statistics pass + b2b {res[2]:.1f} us", flush=True)
