"""Per-layer table of the ResNet-101 train-mode forward at B=128 bf16: every distinct conv geometry, its
multiplicity, st_conv time, and the two floors that bound it (HBM at 8 TB/s, MFMA at 2.5 PFLOP/s), plus the
following bn_act pass.  Debug aid; bench.py is the contract."""
import sys, os
from collections import OrderedDict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
# torchvision ResNet-101 (Bottleneck, [3,4,23,3]), stride on conv2 (v1.5), cnn.py:23-34
geo = OrderedDict()
def add(cin, cout, k, s, p, hh):
    ko = 1 if (k > 1 and cin % 64 == 0) else 0
    key = (cin, cout, k, s, p, ko, hh)
    geo[key] = geo.get(key, 0) + 1
add(3, 64, 7, 2, 3, 224)
hh, inpl = 56, 64
for planes, blocks, stride in ((64, 3, 1), (128, 4, 2), (256, 23, 2), (512, 3, 2)):
    for bi in range(blocks):
        s = stride if bi == 0 else 1
        add(inpl, planes, 1, 1, 0, hh)
        add(planes, planes, 3, s, 1, hh)
        add(planes, planes * 4, 1, 1, 0, hh // s)
        if bi == 0:
            add(inpl, planes * 4, 1, s, 0, hh)
        inpl, hh = planes * 4, hh // s

dt = torch.bfloat16
tot = tot_bn = tot_floor = 0.0
print(f"{'layer':>28} {'x':>3} {'us':>7} {'TF':>5} {'GB/s':>6} {'hbm_us':>7} {'mfma_us':>7} {'eff':>5} | {'bn_us':>6} {'bnGB/s':>6}")
for (cin, cout, k, s, p, ko, hh), mult in geo.items():
    cinp = max(cin, 8)
    x = torch.randn(B, hh, hh, cinp, device="cuda").to(dt)
    w = (torch.randn(cout, k * k * cinp, device="cuda") / (k * k * cinp) ** 0.5).to(dt)
    stats = torch.zeros(2 * cout, device="cuda")
    ho = (hh + 2 * p - k) // s + 1
    cstats = None if os.environ.get("NOSTATS") else stats
    out = ops.conv_nhwc(x, w, k, k, s, p, stats=cstats, k_order=ko)
    y = torch.empty_like(out)
    g = torch.ones(cout, device="cuda"); b = torch.zeros(cout, device="cuda")
    torch.cuda.synchronize()
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    it = 20
    e[0].record()
    for _ in range(it):
        ops.conv_nhwc(x, w, k, k, s, p, stats=cstats, out=out, k_order=ko)
    e[1].record()
    for _ in range(it):
        ops.bn_act(out, g, b, stats=stats, count=float(B * ho * ho), relu=True, out=y)
    e[2].record(); torch.cuda.synchronize()
    us = e[0].elapsed_time(e[1]) / it * 1e3
    bn_us = e[1].elapsed_time(e[2]) / it * 1e3
    flops = 2.0 * B * ho * ho * cout * cin * k * k
    byts = x.numel() * 2 + w.numel() * 2 + out.numel() * 2
    hbm_us = byts / 8e12 * 1e6
    mfma_us = flops / 2.5e15 * 1e6
    fl = max(hbm_us, mfma_us)
    tot += us * mult; tot_bn += bn_us * mult; tot_floor += fl * mult
    print(f"{cin:5d}->{cout:5d} k{k} s{s} @{hh:3d} ko{ko} {mult:3d} {us:7.1f} {flops/us/1e6:5.0f} {byts/us/1e3:6.0f} {hbm_us:7.1f} {mfma_us:7.1f} {fl/us:5.2f} |"
          f" {bn_us:6.1f} {out.numel()*4/bn_us/1e3:6.0f}", flush=True)
print(f"conv total {tot/1e3:.2f} ms, floor {tot_floor/1e3:.2f} ms; bn_act total (no residual) {tot_bn/1e3:.2f} ms")
