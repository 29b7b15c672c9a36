"""A 22-block chain of the layer-3 bottleneck sequence (conv1 1x1 -> bn -> conv2 3x3 -> conv3 1x1 with fused bn2 -> bn3 +
residual), B=128 at 14x14, separate weights per block -- conv kernels judged the way the network runs them (cold filter
banks, freshly written activations), not back to back on one hot problem.  `python tools/chain_bench.py [variant ...]`:
variant 0 = round-1 route (implicit-GEMM conv2 behind a bn_act pass), 1 = image-resident conv2 with bn1 + ReLU in its fill
(st_conv3x3_img), 2 = that plus conv3 on st_conv1x1_wreg (sums conv2's replicated statistics itself), 3 = conv1 on st_conv1x1_kstream, conv2 image-resident,
conv3 on st_conv1x1_astat, separate block-end pass; 4 = that with the block-end pass fused into the next conv1 (st_conv1x1_kfuse:
the four-wave form, measured slower); 5 = the same fusion on the producer / consumer workgroup (st_conv1x1_kfuse8).  Run under `rocprofv3 --kernel-trace --stats` for per-kernel averages."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
from showtell_amd._lib import lib
variants = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 6]
B, h, dt, NL, dev = 128, 14, torch.bfloat16, 22, "cuda"
x0 = torch.relu(torch.randn(B, h, h, 1024, device=dev)).to(dt)
W1 = [(torch.randn(256, 1024, device=dev) / 32).to(dt) for _ in range(NL)]
W2f = [torch.randn(256, 256, 3, 3, device=dev) / 48 for _ in range(NL)]
W2 = [ops.pack_conv_weight(w, dt, k_order=1) for w in W2f]
W2i = [ops.pack_conv_weight_frag(w, ops.conv3x3_img_supported(h, h, 256, 256)) for w in W2f]
W3f = [torch.randn(1024, 256, 1, 1, device=dev) / 16 for _ in range(NL)]
W3 = [w.reshape(1024, 256).to(dt) for w in W3f]
W3i = [ops.pack_conv_weight_frag(w, ops.conv1x1_wreg_supported(256, 1024)) for w in W3f]
W3a = [ops.pack_conv_weight_frag(w, ops.conv1x1_astat_supported(256, 1024)) for w in W3f]
W1f = [w.float().reshape(256, 1024, 1, 1) for w in W1]
W1k = [ops.pack_conv_weight_frag(w, 4) for w in W1f]
s1r = torch.zeros(16, 512, device=dev)
s3r = torch.zeros(4, 2048, device=dev)
g256, b256 = torch.ones(256, device=dev), torch.zeros(256, device=dev)
g1k, b1k = torch.ones(1024, device=dev), torch.zeros(1024, device=dev)
n = float(B * h * h)
y1 = torch.empty(B, h, h, 256, device=dev, dtype=dt); y2 = torch.empty_like(y1); y3 = torch.empty(B, h, h, 1024, device=dev, dtype=dt)
s1, s2, s3 = torch.zeros(512, device=dev), torch.zeros(512, device=dev), torch.zeros(2048, device=dev)
s2r = torch.zeros(16, 512, device=dev)
y3s = [torch.empty(B, h, h, 1024, device=dev, dtype=dt) for _ in range(2)]
xbuf = [torch.empty(B, h, h, 1024, device=dev, dtype=dt) for _ in range(2)]
s3rs = [torch.zeros(4, 2048, device=dev) for _ in range(2)]


def fwd(variant):
    x = x0
    if variant == 6:
        # conv1 (first block only) -> per block: conv2 image-resident, conv3 statistics-only pass, then conv3 + block end + the NEXT block's
        # conv1 in one kernel (st_conv_c3c1); the last block ends with the separate pass
        ops.conv1x1_kstream(x, W1k[0], 256, stats=s1r, stats_replicas=16, out=y1)
        ident = x
        for l in range(NL):
            ops.conv3x3_img(y1, W2i[l], 256, stats=s2r, stats_replicas=16, out=y2, in_bn=dict(stats=s1r, gamma=g256, beta=b256, count=n, replicas=16))
            bn2 = dict(stats=s2r, gamma=g256, beta=b256, count=n, replicas=16)
            if l + 1 < NL:
                ops.conv1x1_astat(y2, W3a[l], 1024, stats=s3rs[l & 1], stats_replicas=4, in_bn=bn2, stats_only=True)
                xn = xbuf[l & 1]
                ops.conv_c3c1(y2, W3a[l], ident, W1k[l + 1], bn2=bn2, bn3=dict(stats=s3rs[l & 1], gamma=g1k, beta=b1k, replicas=4), count=n,
                              stats=s1r, stats_replicas=16, x_out=xn, out=y1)
                ident = xn
            else:
                ops.conv1x1_astat(y2, W3a[l], 1024, stats=s3r, stats_replicas=4, out=y3, in_bn=bn2)
                ops.bn_act(y3, g1k, b1k, stats=s3r, stats_replicas=4, count=n, relu=True, res=ident, out=y3)
        return
    for l in range(NL):
        if variant >= 4:
            # x holds the previous block's output only for l == 0; afterwards (raw3, ident) of the previous block are pending
            if l == 0:
                ops.conv1x1_kstream(x, W1k[l], 256, stats=s1r, stats_replicas=16, out=y1)
                ident = x
            else:
                xn = xbuf[l & 1]
                ops.conv1x1_kfuse(y3s[(l - 1) & 1], ident, W1k[l], dict(stats=s3rs[(l - 1) & 1], gamma=g1k, beta=b1k, count=n, replicas=4),
                                  stats=s1r, stats_replicas=16, x_out=xn, out=y1, eight_waves=(variant == 5))
                ident = xn
            ops.conv3x3_img(y1, W2i[l], 256, stats=s2r, stats_replicas=16, out=y2, in_bn=dict(stats=s1r, gamma=g256, beta=b256, count=n, replicas=16))
            ops.conv1x1_astat(y2, W3a[l], 1024, stats=s3rs[l & 1], stats_replicas=4, out=y3s[l & 1], in_bn=dict(stats=s2r, gamma=g256, beta=b256, count=n, replicas=16))
            if l == NL - 1:
                ops.bn_act(y3s[l & 1], g1k, b1k, stats=s3rs[l & 1], stats_replicas=4, count=n, relu=True, res=ident, out=y3)
            continue
        if variant == 3:
            ops.conv1x1_kstream(x, W1k[l], 256, stats=s1r, stats_replicas=16, out=y1)
            ops.conv3x3_img(y1, W2i[l], 256, stats=s2r, stats_replicas=16, out=y2, in_bn=dict(stats=s1r, gamma=g256, beta=b256, count=n, replicas=16))
            ops.conv1x1_astat(y2, W3a[l], 1024, stats=s3r, stats_replicas=4, out=y3, in_bn=dict(stats=s2r, gamma=g256, beta=b256, count=n, replicas=16))
            ops.bn_act(y3, g1k, b1k, stats=s3r, stats_replicas=4, count=n, relu=True, res=x, out=y3)
            x = y3
            continue
        ops.conv_nhwc(x, W1[l], 1, 1, 1, 0, stats=s1, out=y1)
        if variant == 0:
            ops.bn_act(y1, g256, b256, stats=s1, count=n, relu=True, out=y1)
            ops.conv_nhwc(y1, W2[l], 3, 3, 1, 1, stats=s2, out=y2, k_order=1)
        else:
            ops.conv3x3_img(y1, W2i[l], 256, stats=s2r, stats_replicas=16, out=y2, in_bn=dict(stats=s1, gamma=g256, beta=b256, count=n))
            if variant == 1:
                torch.sum(s2r, 0, out=s2)      # the engine's bn_reduce_replicas launch
        if variant == 2:
            ops.conv1x1_wreg(y2, W3i[l], 1024, stats=s3r, stats_replicas=4, out=y3, in_bn=dict(stats=s2r, gamma=g256, beta=b256, count=n, replicas=16))
            torch.sum(s3r, 0, out=s3)
        else:
            ops.conv_nhwc(y2, W3[l], 1, 1, 1, 0, stats=s3, out=y3, in_bn=dict(stats=s2, gamma=g256, beta=b256, count=n))
        ops.bn_act(y3, g1k, b1k, stats=s3, count=n, relu=True, res=x, out=y3)
        x = y3


for v in variants:
    for _ in range(2):
        fwd(v)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        fwd(v)
    e1.record(); torch.cuda.synchronize()
    print(f"variant {v}: {e0.elapsed_time(e1) / 5 * 1e3 / NL:.1f} us per bottleneck block", flush=True)
