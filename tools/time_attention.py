"""Time the soft-attention training step at BASELINE config 3 (bs=64) and its parts (debug aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import optim
from showtell_amd.cnn_attn import ResNet as ResNetAttn
from showtell_amd.rnn_attn import RNN_Attn
from showtell_amd.train import synthetic_batch
E = H = 512; V = 10000; L = 5
dt = torch.bfloat16
cnn = ResNetAttn(101, E, dtype=dt).cuda().train()
rnn = RNN_Attn(E, 2048, 512, H, V, L, dtype=dt).cuda().train()
opt = optim.SGD(list(rnn.parameters()), lr=0.01, momentum=0.9)
img, cap, lens = synthetic_batch(64, V, seed=5)
def t(f, n=8):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
feat = cnn(img)
print(f"encoder fwd (B=64): {t(lambda: cnn(img)):.2f} ms")
def fb():
    opt.zero_grad(); l = rnn.loss(feat, cap, lens, 1.0); l.backward(); opt.step()
print(f"decoder fwd+bwd+sgd: {t(fb):.2f} ms")
with torch.no_grad():
    print(f"decoder fwd only: {t(lambda: rnn.loss(feat, cap, lens, 1.0)):.2f} ms")
