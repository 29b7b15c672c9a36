"""Per-wave phase timing of st_conv3x3_img (s_memtime stamps written by the kernel through st_debug_stamps):
python tools/img_stamps.py [h c] -> prefetch+fill / barrier wait / K loop / epilogue, in us at the nominal 2.4 GHz tick."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from showtell_amd import ops
from showtell_amd._lib import lib
h, c = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (14, 256)
B = 128
xs = [torch.randn(B, h, h, c, device="cuda").bfloat16() for _ in range(4)]
w = torch.randn(c, c, 3, 3, device="cuda") / (9 * c) ** 0.5
wf = ops.pack_conv_weight_frag(w, ops.conv3x3_img_supported(h, h, c, c))
R = 16
st = torch.zeros(R, 2 * c, device="cuda")
x2 = xs[0].float().reshape(-1, c)
ist = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
g, b = torch.ones(c, device="cuda"), torch.zeros(c, device="cuda")
y = torch.empty(B, h, h, c, device="cuda", dtype=torch.bfloat16)
run = lambda x: ops.conv3x3_img(x, wf, c, stats=st, stats_replicas=R, out=y, in_bn=dict(stats=ist, gamma=g, beta=b, count=float(B * h * h)))
for x in xs:
    run(x)
torch.cuda.synchronize()
buf = torch.zeros(8 * 4 * 8192, dtype=torch.int64, device="cuda")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
run(xs[0])
lib().st_debug_stamps(buf.data_ptr())
e0.record()
run(xs[1])
e1.record()
torch.cuda.synchronize()
lib().st_debug_stamps(None)
print(f"launch: {e0.elapsed_time(e1) * 1e3:.1f} us (HIP events, stamped build path)")
s = buf.cpu().numpy().reshape(-1, 8)
s = s[s[:, 0] != 0].astype(np.float64)
d = np.diff(s[:, :5], axis=1) / 2400.0
print(f"{len(s)} waves")
for i, nm in enumerate(["weight prefetch + fill", "barrier wait", "K loop", "epilogue"]):
    print(f"  {nm:>24}: mean {d[:, i].mean():6.2f} us  p10 {np.percentile(d[:, i], 10):6.2f}  p90 {np.percentile(d[:, i], 90):6.2f}")
print(f"  {'wave total':>24}: mean {d.sum(1).mean():6.2f} us")
