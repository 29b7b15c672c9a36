"""Fused vocabulary projection + cross entropy (csrc/vocab_ce.hip) against the launch chain (st_rnn_forward's logits + st_cross_entropy) on
bench.py's decoder shape: loss, every gradient, and the time of forward + backward.  usage: python tools/check_fused_ce.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.rnn import RNN
from showtell_amd.train import synthetic_batch
dev = "cuda"
E, H, L, V, B = 512, 512, 5, 10000, 128
res = {}
for fused in ("0", "1"):
    os.environ["ST_FUSED_CE"] = fused
    torch.manual_seed(1)
    rnn = RNN(E, H, V, L, dtype=torch.bfloat16).to(dev).train()
    _, caption, lens = synthetic_batch(B, V, seed=1, device=dev)
    feat = torch.randn(B, E, device=dev, generator=torch.Generator(device=dev).manual_seed(5)).requires_grad_(True)
    loss = rnn.loss(feat, caption, lens)
    loss.backward()
    torch.cuda.synchronize()
    g = {k: p.grad.detach().float().clone() for k, p in rnn.named_parameters() if p.grad is not None}
    g["feat"] = feat.grad.detach().float().clone()
    res[fused] = (float(loss), g)
    for _ in range(3):
        rnn.zero_grad(); l2 = rnn.loss(feat, caption, lens); l2.backward()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        l2 = rnn.loss(feat, caption, lens); l2.backward()
    e1.record(); torch.cuda.synchronize()
    print(f"ST_FUSED_CE={fused}: loss {float(loss):.6f}   decoder forward + loss + backward {e0.elapsed_time(e1) / 10:.3f} ms", flush=True)
l0, g0 = res["0"]; l1, g1 = res["1"]
print(f"loss difference {abs(l0 - l1):.3e}")
worst = 0.0
for k in g0:
    d = (g0[k] - g1[k]).abs().max().item(); s = g0[k].abs().max().item()
    worst = max(worst, d / max(s, 1e-12))
    print(f"  {k:32s} max|diff| {d:.3e}  max|grad| {s:.3e}  rel {d / max(s, 1e-12):.3e}")
print("worst relative difference", worst)
