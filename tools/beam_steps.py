"""One beam-5 search over 256 images' features at bench.py's decoder shape (the program to trace with rocprofv3 --kernel-trace --stats)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.rnn import RNN
torch.manual_seed(3)
rnn = RNN(512, 512, 10000, 5, dtype=torch.bfloat16).cuda().eval()
f = torch.randn(256, 512, device="cuda")
for _ in range(3):
    rnn.beam_search(f, 5, 1, 25)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); rnn.beam_search(f, 5, 1, 25); e1.record(); torch.cuda.synchronize()
print(f"beam-5 over 256 images, 25 iterations: {e0.elapsed_time(e1):.2f} ms")
