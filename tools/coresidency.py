"""Can an HBM-bound pass (st_bn_act at the layer3 size) run UNDER an MFMA kernel of another stream?  Times N launches of each alone
and both streams together (debug aid for the pipelined schedule: DESIGN.md 4b)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops

B, N = 128, 40
x256 = torch.randn(B, 14, 14, 256, device="cuda").bfloat16()
x1024 = torch.randn(B, 14, 14, 1024, device="cuda").bfloat16()
ident = torch.randn(B, 14, 14, 1024, device="cuda").bfloat16()
raw = torch.randn(B, 14, 14, 1024, device="cuda").bfloat16()
w3 = ops.pack_conv_weight_frag(torch.randn(1024, 256, 1, 1, device="cuda") / 16, ops.conv1x1_astat_supported(256, 1024))
w1 = ops.pack_conv_weight_frag(torch.randn(256, 1024, 1, 1, device="cuda") / 32, 4)
w2 = ops.pack_conv_weight_frag(torch.randn(256, 256, 3, 3, device="cuda") / 48, ops.conv3x3_img_supported(14, 14, 256, 256))
st1024 = torch.zeros(4, 2048, device="cuda"); st256 = torch.zeros(4, 512, device="cuda")
y1024 = torch.empty_like(x1024); y256 = torch.empty_like(x256); yb = torch.empty_like(raw)
gam, bet = torch.ones(1024, device="cuda"), torch.zeros(1024, device="cuda")
stats = torch.zeros(2048, device="cuda"); stats[1024:] = 25088.0
kern = {
    "astat": lambda: ops.conv1x1_astat(x256, w3, 1024, stats=st1024, stats_replicas=4, out=y1024),
    "kstream": lambda: ops.conv1x1_kstream(x1024, w1, 256, stats=st256, stats_replicas=4, out=y256),
    "img": lambda: ops.conv3x3_img(x256, w2, 256, stats=st256, stats_replicas=4, out=y256),
    "bn_act": lambda: ops.bn_act(raw, gam, bet, stats=stats, count=25088.0, relu=True, res=ident, out=yb),
}
sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def run(fa, fb):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    sa.wait_event(e0); sb.wait_event(e0)
    for _ in range(N):
        if fa:
            with torch.cuda.stream(sa):
                fa()
        if fb:
            with torch.cuda.stream(sb):
                fb()
    torch.cuda.current_stream().wait_stream(sa); torch.cuda.current_stream().wait_stream(sb)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


for f in kern.values():
    for _ in range(3):
        f()
alone = {k: run(f, None) for k, f in kern.items()}
print("alone (us per launch):", {k: round(v, 1) for k, v in alone.items()}, flush=True)
for a in ("astat", "kstream", "img"):
    for b in ("bn_act", "astat", "kstream", "img"):
        t = run(kern[a], kern[b])
        print(f"{a:8s} || {b:8s}: {t:6.1f} us per pair  (sum alone {alone[a] + alone[b]:6.1f}, max {max(alone[a], alone[b]):6.1f})", flush=True)
