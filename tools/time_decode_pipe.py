"""Greedy decode at the BASELINE shape, launch chain vs the pipelined (layer-per-XCD) decoder, same process (ST_DECODE_PIPE switch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.rnn import RNN
E, H, V, L = 512, 512, 10000, 5
m = RNN(E, H, V, L, dtype=torch.bfloat16).cuda().eval()
for B in (128, 256, 32):
    feat = torch.randn(B, E, device="cuda")
    res = {}
    for mode in ("0", "1"):
        os.environ["ST_DECODE_PIPE"] = mode
        for _ in range(3):
            ids = m.sentence_index(feat)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 20
        e0.record()
        for _ in range(n):
            ids = m.sentence_index(feat)
        e1.record(); torch.cuda.synchronize()
        res[mode] = (e0.elapsed_time(e1) / n, ids.clone())
    byts = (13009680 + 2 * L * B * H + B * E) * 2 + 16 * B
    for mode, nm in (("0", "launch chain"), ("1", "pipelined   ")):
        ms = res[mode][0]
        print(f"B={B:4d} {nm}: {ms:.3f} ms / 25 steps = {ms / 25 * 1e3:6.1f} us/step  {byts / (ms / 25 * 1e3) / 1e6:.3f} TB/s = {byts / (ms / 25 * 1e3) / 1e6 / 8 * 100:4.1f} % of 8 TB/s  {B / ms * 1e3:8.0f} captions/s")
    print("      ids equal:", bool(torch.equal(res["0"][1], res["1"][1])))
