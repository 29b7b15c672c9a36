import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn
from tests._util import load_fixture
from tests.test_gpu_decoder import _make, _rel
cell, name = sys.argv[1], sys.argv[2]
params, grads, d = load_fixture(name)
m = _make(cell, params, torch.float32)
feat = torch.from_numpy(d["feat"]).cuda().requires_grad_(True)
cap, lens = torch.from_numpy(d["caption"]).cuda(), d["lens"].tolist()
loss = m.loss(feat, cap, lens)
print("loss", loss.item(), float(d["loss"]))
loss.backward()
for k, g in grads.items():
    p = dict(m.named_parameters())[k]
    print(k, "%.3e" % _rel(p.grad, g), float(p.grad.abs().max()), float(g.abs().max()))
