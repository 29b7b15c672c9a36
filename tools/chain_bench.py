"""A 22-block chain of the layer-3 bottleneck sequence (conv1 1x1 -> bn -> conv2 3x3 -> conv3 1x1 with fused bn2 -> bn3 +
residual), B=128 at 14x14, separate weights per block -- conv kernels judged the way the network runs them (cold filter
banks, freshly written activations), not back to back on one hot problem.  `python tools/chain_bench.py [ring]` with
ring = st_tune's first knob (0 default dispatch, 3 = experimental small-block kernel on the 3x3 layers); run it under
`rocprofv3 --kernel-trace --stats` for per-kernel averages."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
from showtell_amd._lib import lib
rings = [int(a) for a in sys.argv[1:]] or [0, 3]
B, h, dt, NL, dev = 128, 14, torch.bfloat16, 22, "cuda"
x0 = torch.relu(torch.randn(B, h, h, 1024, device=dev)).to(dt)
W1 = [(torch.randn(256, 1024, device=dev) / 32).to(dt) for _ in range(NL)]
W2 = [(torch.randn(256, 9 * 256, device=dev) / 48).to(dt) for _ in range(NL)]
W3 = [(torch.randn(1024, 256, device=dev) / 16).to(dt) for _ in range(NL)]
g256, b256 = torch.ones(256, device=dev), torch.zeros(256, device=dev)
g1k, b1k = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
n = float(B * h * h)
y1 = torch.empty(B, h, h, 256, device=dev, dtype=dt); y2 = torch.empty_like(y1); y3 = torch.empty(B, h, h, 1024, device=dev, dtype=dt)
s1, s2, s3 = torch.zeros(512, device=dev), torch.zeros(512, device=dev), torch.zeros(2048, device=dev)


def fwd():
    x = x0
    for l in range(NL):
        ops.conv_nhwc(x, W1[l], 1, 1, 1, 0, stats=s1, out=y1)
        ops.bn_act(y1, g256, b256, stats=s1, count=n, relu=True, out=y1)
        ops.conv_nhwc(y1, W2[l], 3, 3, 1, 1, stats=s2, out=y2, k_order=1)
        ops.conv_nhwc(y2, W3[l], 1, 1, 1, 0, stats=s3, out=y3, in_bn=dict(stats=s2, gamma=g256, beta=b256, count=n))
        ops.bn_act(y3, g1k, b1k, stats=s3, count=n, relu=True, res=x, out=y3)
        x = y3


for ring in rings:
    lib().st_tune(ring, 0, 1)
    for _ in range(2):
        fwd()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fwd()
    e1.record(); torch.cuda.synchronize()
    print(f"ring {ring}: {e0.elapsed_time(e1) / 5 * 1e3 / NL:.1f} us per bottleneck block", flush=True)
lib().st_tune(0, 0, 1)
