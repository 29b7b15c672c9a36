"""Per-wave phase timing of st_conv_c3c1 (s_memtime stamps through st_debug_stamps), 256 -> 1024 -> 256 at 14 x 14, B = 128.
usage: python tools/c3_stamps.py [train|eval]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from showtell_amd import ops
from showtell_amd._lib import lib
mode = sys.argv[1] if len(sys.argv) > 1 else "train"
B, h = 128, 14
n = float(B * h * h)
x2s = [torch.randn(B, h, h, 256, device="cuda").bfloat16() for _ in range(3)]
ids = [torch.relu(torch.randn(B, h, h, 1024, device="cuda")).bfloat16() for _ in range(3)]
w3 = ops.pack_conv_weight_frag(torch.randn(1024, 256, 1, 1, device="cuda") / 16, 2)
w1 = ops.pack_conv_weight_frag(torch.randn(256, 1024, 1, 1, device="cuda") / 32, 4)
g256, b256 = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
g1k, b1k = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
x2f = x2s[0].float().reshape(-1, 256)
s2 = torch.zeros(4, 512, device="cuda"); s2[0] = torch.cat([x2f.sum(0), (x2f * x2f).sum(0)])
bn2 = dict(stats=s2, gamma=g256, beta=b256, count=n, replicas=4)
s3 = torch.zeros(4, 2048, device="cuda")
ops.conv1x1_astat(x2s[0], w3, 1024, stats=s3, stats_replicas=4, in_bn=bn2, stats_only=True)
s1 = torch.zeros(4, 512, device="cuda")
xo = torch.empty(B, h, h, 1024, device="cuda", dtype=torch.bfloat16); y = torch.empty(B, h, h, 256, device="cuda", dtype=torch.bfloat16)
if mode == "train":
    run = lambda i: ops.conv_c3c1(x2s[i], w3, ids[i], w1, bn2=bn2, bn3=dict(stats=s3, gamma=g1k, beta=b1k, replicas=4), count=n, stats=s1, stats_replicas=4, x_out=xo, out=y)
else:
    run = lambda i: ops.conv_c3c1(x2s[i], w3, ids[i], w1, scale3=g1k, shift3=b1k, scale1=g256, shift1=b256, x_out=xo, out=y)
for i in range(3): run(i)
torch.cuda.synchronize()
bufs = [torch.zeros(16 * 4 * 1024, dtype=torch.int64, device="cuda") for _ in range(3)]
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(3):
    lib().st_debug_stamps(bufs[i].data_ptr()); run(i)
e1.record()
torch.cuda.synchronize()
lib().st_debug_stamps(None)
print(f"{mode}: 3 launches: {e0.elapsed_time(e1) * 1e3 / 3:.1f} us each")
s = bufs[1].cpu().numpy().reshape(-1, 16)
s = s[s[:, 0] != 0].astype(np.float64)
d = np.diff(s[:, :13], axis=1) / 2400.0
names = ["prefetch + fill", "fill barrier"] + [f"iteration {c} (A({c})" + (f" + B({c-1})/E({c}))" if c else " + E(0))") for c in range(8)] + ["B(7)", "conv1 epilogue + statistics"]
for i, nm in enumerate(names):
    print(f"  {nm:>32}: mean {d[:, i].mean():6.2f} us  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f}   (s_memtime / 2400: nominal-clock us)")
print(f"  wave total {d.sum(1).mean():.2f} us over {len(s)} waves")
t0 = None
for i, b_ in enumerate(bufs):
    q = b_.cpu().numpy().reshape(-1, 16); q = q[q[:, 0] != 0].astype(np.float64)
    ent, ext = q[:, 14] * 0.01, q[:, 15] * 0.01
    t0 = ent.min() if t0 is None else t0
    print(f"  launch {i}: first entry {ent.min()-t0:7.2f}  last entry {ent.max()-t0:7.2f}  first exit {ext.min()-t0:7.2f}  median exit {np.median(ext)-t0:7.2f}  last exit {ext.max()-t0:7.2f} us")
