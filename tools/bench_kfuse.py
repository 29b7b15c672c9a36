"""Block-end normalise pass + conv1 (st_bn_act, st_conv1x1_wreg) against the fused loader (st_conv1x1_kfuse) at the layer1 / layer2
shapes of ResNet-101, B = 128 (debug aid; bench.py is the contract)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops

B = 128
for (H, C, N, idbn) in [(56, 256, 64, False), (56, 256, 64, True), (56, 256, 128, False), (28, 512, 128, False), (28, 512, 128, True), (28, 512, 256, False)]:
    ntw = ops.conv1x1_kfuse_supported(C, N)
    raw = torch.randn(B, H, H, C, device="cuda").bfloat16()
    ident = torch.randn(B, H, H, C, device="cuda").bfloat16()
    w = torch.randn(N, C, 1, 1, device="cuda") / C ** 0.5
    wf = ops.pack_conv_weight_frag(w, ntw)
    gam, bet = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    r2 = raw.float().reshape(-1, C)
    st = torch.zeros(4, 2 * C, device="cuda"); st[0] = torch.cat([r2.sum(0), (r2 * r2).sum(0)])
    n = float(B * H * H)
    so = torch.zeros(4, 2 * N, device="cuda")
    x = torch.empty_like(raw); y = torch.empty(B, H, H, N, device="cuda", dtype=torch.bfloat16)
    bn = dict(stats=st, gamma=gam, beta=bet, count=n, replicas=4)
    idb = dict(stats=st, gamma=gam, beta=bet, replicas=4) if idbn else None
    rb = dict(stats=st, gamma=gam, beta=bet, stats_replicas=4) if idbn else None

    def sep():
        ops.bn_act(raw, gam, bet, stats=st, count=n, relu=True, res=ident, res_bn=rb, out=x, stats_replicas=4)
        ops.conv1x1_wreg(x, wf, N, stats=so, stats_replicas=4, out=y)

    def fused():
        ops.conv1x1_kfuse(raw, ident, wf, bn, N=N, id_bn=idb, stats=so, stats_replicas=4, x_out=x, out=y)

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
    gb = (3 * raw.numel() * 2 + y.numel() * 2) / 1e9
    print(f"{H}x{H} {C}->{N} id_bn={idbn}: bn_act + wreg {res[0]:.1f} us, fused {res[1]:.1f} us ({gb / res[1] * 1e6 / 1e3:.2f} TB/s of {gb:.2f} GB)", flush=True)
