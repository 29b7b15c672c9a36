"""Time greedy / beam decoding at the BASELINE shapes (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.rnn import RNN
E, H, V, L = 512, 512, 10000, 5
for dtype in (torch.bfloat16, torch.float32):
    m = RNN(E, H, V, L, dtype=dtype).cuda().eval()
    B = 128
    feat = torch.randn(B, E, device="cuda")
    for _ in range(3):
        m.sentence_index(feat)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        m.sentence_index(feat)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    per_step = ms / 25 * 1e3
    byts = (13009680 + 2 * L * B * H + B * E) * (2 if dtype == torch.bfloat16 else 4) + 16 * B
    print(f"greedy {dtype} B={B}: {ms:.3f} ms / 25 steps = {per_step:.1f} us/step; algorithmic {byts/1e6:.2f} MB/step -> {byts/per_step/1e6:.3f} TB/s = {byts/per_step/1e6/8*100:.1f}% of 8 TB/s; {B/ms*1e3:.0f} captions/s")
m = RNN(E, H, V, L, dtype=torch.bfloat16).cuda().eval()
with torch.no_grad():
    m.linear.weight *= 12.0; m.linear.bias[2] += 1.5
feat = torch.randn(256, E, device="cuda")
m.beam_search(feat, 5, 1, 25)
torch.cuda.synchronize(); t0 = time.time()
out = m.beam_search(feat, 5, 1, 25)
torch.cuda.synchronize(); dt = time.time() - t0
print(f"beam=5 B=256 max_length=25 bf16: {dt*1e3:.1f} ms -> {256/dt:.0f} captions/s; non-empty {sum(1 for o in out if o)}")
