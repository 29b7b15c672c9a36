"""Per-block phase timing of st_conv launches (stamps written by the kernel, st_debug_stamps):
python tools/conv_stamps.py cin cout k s h   -> where a block's time goes (s_memtime, shader cycles, in-block differences)
and how the blocks of two back-to-back launches sit on the chip-wide 100 MHz clock (s_memrealtime)."""
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
R = 16
stats = torch.zeros(R, 2 * cout, device="cuda")
KO = 1 if (k > 1 and cin % 64 == 0) else 0
run = lambda x, out=None: ops.conv_nhwc(x, w, k, k, s, k // 2, stats=stats, out=out, k_order=KO, stats_replicas=R)
for i in range(6):
    out = run(xs[i])
torch.cuda.synchronize()
bufs = [torch.zeros(8 * 65536, dtype=torch.int64, device="cuda") for _ in range(3)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(3):                       # three back-to-back launches, each with its own stamp buffer
    lib().st_debug_stamps(bufs[i].data_ptr())
    run(xs[i], out)
e1.record()
torch.cuda.synchronize()
lib().st_debug_stamps(None)
print(f"3 launches: {e0.elapsed_time(e1) * 1e3 / 3:.1f} us per launch (HIP events)")
sts = [b.cpu().numpy().reshape(-1, 8) for b in bufs]
nb = int((sts[0][:, 0] != 0).sum())
st = sts[1][:nb].astype(np.float64)
tick = 1.0 / 2400.0                      # s_memtime: shader cycles
d = np.diff(st[:, :5], axis=1) * tick
names = ["addr set-up", "first tile in LDS", "K loop", "epilogue"]
print(f"{nb} blocks")
for i in range(4):
    print(f"  {names[i]:>18}: mean {d[:, i].mean():6.2f} us  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f}")
print(f"  {'block total':>18}: mean {d.sum(1).mean():6.2f} us")
# chip-wide timeline (10 ns ticks)
ent = [x[:nb, 6].astype(np.float64) * 0.01 for x in sts]
ext = [x[:nb, 7].astype(np.float64) * 0.01 for x in sts]
t0 = ent[0].min()
for i in range(3):
    print(f"  launch {i}: first entry {ent[i].min()-t0:7.2f}  median entry {np.median(ent[i])-t0:7.2f}  last entry {ent[i].max()-t0:7.2f}  "
          f"first exit {ext[i].min()-t0:7.2f}  last exit {ext[i].max()-t0:7.2f} us")
print(f"  period (first entry to first entry): {ent[2].min()-ent[1].min():.2f} us; gap last exit -> next first entry: {ent[2].min()-ext[1].max():.2f} us")
hw = sts[1][:nb, 5].astype(np.int64)
cu = ((hw >> 32) & 0xf) * 10000 + ((hw >> 13) & 0x7) * 100 + ((hw >> 8) & 0xf)
u, c = np.unique(cu, return_counts=True)
print(f"  distinct (xcc,se,cu) keys: {len(u)}; blocks per key min {c.min()} max {c.max()}")
one = cu == cu[0]
order = np.argsort(ent[1][one])
print("  one CU's blocks (entry, exit) us:", [(round(float(a - t0), 1), round(float(b - t0), 1)) for a, b in zip(ent[1][one][order], ext[1][one][order])][:12])
ts = np.linspace(ent[1].min(), ext[1].max(), 12)[1:-1]
print("  blocks in flight over the launch:", [int(((ent[1] <= t) & (ext[1] > t)).sum()) for t in ts])
