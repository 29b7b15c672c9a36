"""Per-block phase timing of one st_conv launch (s_memtime stamps written by the kernel, st_debug_stamps):
python tools/conv_stamps.py cin cout k s h   -> where a block's time goes and how the blocks are spread over time."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from showtell_amd import ops
from showtell_amd._lib import lib
cin, cout, k, s, h = map(int, sys.argv[1:6])
B = 128
dt = torch.bfloat16
xs = [torch.randn(B, h, h, cin, device="cuda").to(dt) for _ in range(6)]
w = (torch.randn(cout, k * k * cin, device="cuda") / (k * k * cin) ** 0.5).to(dt)
stats = torch.zeros(2 * cout, device="cuda")
KO = 1 if (k > 1 and cin % 64 == 0) else 0
for i in range(6):
    out = ops.conv_nhwc(xs[i], w, k, k, s, k // 2, stats=stats, k_order=KO)
torch.cuda.synchronize()
buf = torch.zeros(8 * 65536, dtype=torch.int64, device="cuda")
lib().st_debug_stamps(buf.data_ptr())
ops.conv_nhwc(xs[0], w, k, k, s, k // 2, stats=stats, out=out, k_order=KO)
torch.cuda.synchronize()
lib().st_debug_stamps(None)
st = buf.cpu().numpy().reshape(-1, 8)
nb = int((st[:, 0] != 0).sum())
st = st[:nb].astype(np.float64)
t0 = st[:, 0].min()
tick = 1.0 / 2400.0   # s_memtime counts shader cycles (~2.4 GHz): us per tick; differences inside one block only
rel = (st[:, :5] - t0) * tick
print(f"{nb} blocks (counters of different XCDs are not aligned: only in-block differences are meaningful)")
names = ["entry", "addr set-up", "first tile in LDS", "K loop", "epilogue"]
d = np.diff(rel, axis=1)
for i in range(4):
    print(f"  {names[i+1]:>18}: mean {d[:, i].mean():6.2f} us  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f}")
print(f"  {'block total':>18}: mean {(rel[:, 4]-rel[:, 0]).mean():6.2f} us")
e = (st[:, [6, 7, 4]] - st[:, [3, 6, 7]]) * tick
for i, nm in enumerate(["epi: stats", "epi: first half", "epi: second half"]):
    print(f"  {nm:>18}: mean {e[:, i].mean():6.2f} us  p10 {np.percentile(e[:, i], 10):6.2f}  p90 {np.percentile(e[:, i], 90):6.2f}")
hw = st[:, 5].astype(np.int64)
cu = ((hw >> 32) & 0xf) * 1000 + ((hw >> 13) & 0x3) * 100 + ((hw >> 8) & 0xf)   # (xcc, se, cu) key
u, c = np.unique(cu, return_counts=True)
print(f"  distinct (xcc,se,cu): {len(u)}; blocks per CU min {c.min()} max {c.max()}")
