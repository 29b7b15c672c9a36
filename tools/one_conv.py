"""Run one conv shape repeatedly (for rocprofv3 --pmc): python tools/one_conv.py cin cout k s h ring kc w8"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
from showtell_amd._lib import lib
cin, cout, k, s, h, ring, kc, w8 = map(int, sys.argv[1:9])
B = 128
x = torch.randn(B, h, h, cin, device="cuda").to(torch.bfloat16)
w = (torch.randn(cout, k * k * cin, device="cuda") / (k * k * cin) ** 0.5).to(torch.bfloat16)
stats = torch.zeros(2 * cout, device="cuda")
lib().st_tune(ring, kc, w8)
KO = 1 if (k > 1 and cin % 64 == 0) else 0
out = None
for _ in range(5):
    out = ops.conv_nhwc(x, w, k, k, s, k // 2, stats=stats, out=out, k_order=KO)
torch.cuda.synchronize()
