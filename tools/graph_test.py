"""Do HIP graphs shorten the dependent-launch chains?  Encoder forward (144 launches) and greedy decode (150 launches), eager vs
captured with torch.cuda.CUDAGraph (hipGraph) on the same stream (debug aid)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd.cnn import ResNet
from showtell_amd.rnn import RNN

dev = torch.device("cuda", 0)


def timeit(fn, n):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


what = sys.argv[1] if len(sys.argv) > 1 else "both"
if what in ("enc", "both"):
    m = ResNet(101, 512, dtype=torch.bfloat16).to(dev).train()
    x = torch.randn(128, 3, 224, 224, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            m.backbone_features(x)
        torch.cuda.synchronize()
        t_e = timeit(lambda: m.backbone_features(x), 10)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            out = m.backbone_features(x)
        t_g = timeit(g.replay, 10)
    print(f"encoder forward: eager {t_e:.3f} ms, graph replay {t_g:.3f} ms", flush=True)
if what in ("dec", "both"):
    r = RNN(512, 512, 10000, 5, dtype=torch.bfloat16).to(dev).eval()
    feat = torch.randn(128, 512, device=dev)
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            r.sentence_index(feat)
        torch.cuda.synchronize()
        t_e = timeit(lambda: r.sentence_index(feat), 10)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            ids = r.sentence_index(feat)
        t_g = timeit(g.replay, 10)
    print(f"greedy decode (25 steps): eager {t_e * 1e3 / 25:.1f} us/step, graph replay {t_g * 1e3 / 25:.1f} us/step", flush=True)
