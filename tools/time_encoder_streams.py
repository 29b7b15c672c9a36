"""Backbone forwards only (no trainable part), issued round-robin on NS streams: what the software-pipelined step could reach if the
trainable work cost nothing.  usage: python tools/time_encoder_streams.py [NS ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.cnn import ResNet
B = 128
m = ResNet(101, 512, dtype=torch.bfloat16).cuda().train()
x = torch.randn(B, 3, 224, 224, device="cuda")
for ns in [int(a) for a in sys.argv[1:]] or [1, 2, 3]:
    streams = [torch.cuda.Stream() for _ in range(ns)]
    def run(n):
        for k in range(n):
            with torch.cuda.stream(streams[k % ns]):
                m.backbone_features(x)
    run(6); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for s in streams: s.wait_event(e0)
    run(n)
    for s in streams: torch.cuda.current_stream().wait_stream(s)
    e1.record(); torch.cuda.synchronize()
    print(f"{ns} stream(s): {e0.elapsed_time(e1) / n:.3f} ms per forward  ({B * n / e0.elapsed_time(e1) * 1e3:.0f} img/s)", flush=True)
