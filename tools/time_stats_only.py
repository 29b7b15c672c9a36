"""Statistics-only conv3 passes (train mode; the outputs are recomputed by st_conv_c3c1 / st_conv_b2b) at B = 128 against the number of statistics
replicas: the passes do not depend on it (the per-workgroup atomics are not what bounds them).  usage: python tools/time_stats_only.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops
B=128
for (C,N,H) in ((128,512,28),(64,256,56),(256,1024,14)):
    x = torch.randn(B,H,H,C, device="cuda").bfloat16()
    w = (torch.randn(N,C,1,1, device="cuda")/C**0.5)
    f = ops.conv1x1_astat if C==256 else ops.conv1x1_wreg
    ntw = ops.conv1x1_astat_supported(C,N) if C==256 else ops.conv1x1_wreg_supported(C,N)
    wf = ops.pack_conv_weight_frag(w, ntw)
    x2 = x.float().reshape(-1,C); ist = torch.cat([x2.sum(0),(x2*x2).sum(0)]).contiguous()
    g,b = torch.ones(C,device="cuda"), torch.zeros(C,device="cuda")
    for R in (1,4,16,64):
        st = torch.zeros(R, 2*N, device="cuda")
        run = lambda: f(x, wf, N, stats=st, stats_replicas=R, stats_only=True, in_bn=dict(stats=ist,gamma=g,beta=b,count=float(B*H*H)))
        for _ in range(5): run()
        torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): run()
        e1.record(); torch.cuda.synchronize()
        print(f"{C}->{N} @{H}: replicas {R:3d}: {e0.elapsed_time(e1)/50*1e3:.1f} us", flush=True)
