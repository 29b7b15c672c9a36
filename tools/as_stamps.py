"""Per-wave phase timing of st_conv1x1_astat (s_memtime stamps through st_debug_stamps), 256 -> 1024 at 14 x 14, B = 128."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from showtell_amd import ops
from showtell_amd._lib import lib
B, h, c, n = 128, 14, 256, 1024
STATS_ONLY = len(sys.argv) > 1 and sys.argv[1] == "stats"      # the statistics-only pass (y == NULL) of the train-mode 14 x 14 conv3
xs = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(3)]
w = torch.randn(n, c, 1, 1, device="cuda") / c ** 0.5
wf = ops.pack_conv_weight_frag(w, ops.conv1x1_astat_supported(c, n))
st = torch.zeros(4, 2 * n, device="cuda")
x2 = xs[0].float().reshape(-1, c)
ist = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
g_, b_ = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
y = torch.empty(B, h, h, n, device="cuda", dtype=torch.bfloat16)
run = lambda x: ops.conv1x1_astat(x, wf, n, stats=st, stats_replicas=4, out=None if STATS_ONLY else y, stats_only=STATS_ONLY,
                                  in_bn=dict(stats=ist, gamma=g_, beta=b_, count=float(B * h * h)))
for x in xs: run(x)
torch.cuda.synchronize()
buf = torch.zeros(8 * 4 * 4096, dtype=torch.int64, device="cuda")
lib().st_debug_stamps(buf.data_ptr())
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
bufs = [torch.zeros(8 * 4 * 4096, dtype=torch.int64, device="cuda") for _ in range(3)]
e0.record()
for i in range(3):
    lib().st_debug_stamps(bufs[i].data_ptr()); run(xs[i])
e1.record()
torch.cuda.synchronize()
buf = bufs[1]
lib().st_debug_stamps(None)
print(f"3 launches: {e0.elapsed_time(e1) * 1e3 / 3:.1f} us each")
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0].astype(np.float64)
d = np.diff(s[:, :6], axis=1) / 2400.0
for i, nm in enumerate(["filter prefetch + fill", "chunk 0", "first half of the chunks", "second half + last epilogue", "statistics flush"]):
    print(f"  {nm:>40}: mean {d[:, i].mean():6.2f} us  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f}")
print(f"  wave total {d.sum(1).mean():.2f} us over {len(s)} waves")

t0 = None
for i, b_ in enumerate(bufs):
    q = b_.cpu().numpy().reshape(-1, 8); q = q[q[:, 0] != 0].astype(np.float64)
    ent, ext = q[:, 6] * 0.01, q[:, 7] * 0.01          # 100 MHz chip-wide clock -> us
    t0 = ent.min() if t0 is None else t0
    print(f"  launch {i}: first entry {ent.min()-t0:7.2f}  median entry {np.median(ent)-t0:7.2f}  last entry {ent.max()-t0:7.2f}  first exit {ext.min()-t0:7.2f}  median exit {np.median(ext)-t0:7.2f}  last exit {ext.max()-t0:7.2f} us")
