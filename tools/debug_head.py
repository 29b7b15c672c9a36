import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn as nn, copy
from showtell_amd.head import linear_bn1d
g = torch.Generator().manual_seed(8)
B, F, E = 32, 256, 64
lin, bn = nn.Linear(F, E), nn.BatchNorm1d(E, momentum=0.01)
x = torch.randn(B, F, generator=g)
lin_d, bn_d = copy.deepcopy(lin).cuda(), copy.deepcopy(bn).cuda()
y_ref = bn(lin(x)); w = torch.randn(B, E, generator=g)
(y_ref * w).sum().backward()
y = linear_bn1d(x.cuda(), lin_d, bn_d, True, torch.float32)
(y * w.cuda()).sum().backward()
for n, a, b in (("w", lin_d.weight, lin.weight), ("b", lin_d.bias, lin.bias), ("g", bn_d.weight, bn.weight), ("beta", bn_d.bias, bn.bias)):
    print(n, (a.grad.cpu() - b.grad).abs().max().item(), b.grad.abs().max().item(), a.grad.abs().max().item())
