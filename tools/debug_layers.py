"""Layer-by-layer error growth of the bf16 train-mode backbone vs the fp32 oracle (debug aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from oracle import restatement as R
from showtell_amd import ops

version, B, size = 50, 4, 128
dtype = torch.bfloat16
params = R.init_encoder_params(version, 64, seed=1)
x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(5))

def rel(got, ref):
    got = got.float().cpu(); ref = ref.float()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-6)).item(), ((got-ref).pow(2).mean().sqrt()/ref.pow(2).mean().sqrt()).item()

def emu(t):  # bf16 storage emulation
    return t.bfloat16().float()

convs = R.resnet_conv_list(version)
def dev_conv(xd, key, k, s, p, cin_pad=None):
    w = params[key + ".weight"].cuda()
    wd = ops.pack_conv_weight(w, dtype, cin_pad)
    C = w.shape[0]
    st = torch.zeros(2 * C, device="cuda")
    y = ops.conv_nhwc(xd, wd, k, k, s, p, stats=st)
    return y, st

# device path
xd = ops.nchw_to_nhwc(x.cuda(), dtype, 8)
# oracle fp32 and emulated
xo = x.clone(); xe = emu(x)
def obn(t, bn, relu=True, train=True):
    y = F.batch_norm(t, None, None, params[bn + ".weight"], params[bn + ".bias"], True, 0.1, 1e-5)
    return F.relu(y) if relu else y
def ebn(t, bn):  # emulate: stats from fp32 conv output, apply on bf16-rounded raw
    m = t.mean((0, 2, 3), keepdim=True); v = t.var((0, 2, 3), unbiased=False, keepdim=True)
    g = params[bn + ".weight"].view(1, -1, 1, 1); b = params[bn + ".bias"].view(1, -1, 1, 1)
    sc = g * torch.rsqrt(v + 1e-5); sh = b - m * sc
    return emu(t) * sc + sh

y, st = dev_conv(xd, "model.0", 7, 2, 3, 8)
yo = F.conv2d(xo, params["model.0.weight"], None, 2, 3)
ye = F.conv2d(xe, emu(params["model.0.weight"]), None, 2, 3)
print("stem raw", rel(y.permute(0,3,1,2), yo), "emu", rel(y.permute(0,3,1,2), ye))
n = y.numel() // 64
a = ops.bn_act(y, params["model.1.weight"].cuda(), params["model.1.bias"].cuda(), stats=st, count=n, relu=True)
ao = obn(yo, "model.1"); ae = emu(F.relu(ebn(ye, "model.1")))
print("stem act", rel(a.permute(0,3,1,2), ao), "emu", rel(a.permute(0,3,1,2), ae))
a = ops.maxpool3x3s2(a); ao = F.max_pool2d(ao, 3, 2, 1); ae = F.max_pool2d(ae, 3, 2, 1)
kind, blocks = R.RESNET_SPECS[version]
for li, nb in enumerate(blocks):
    for bi in range(nb):
        p = f"model.{4+li}.{bi}"
        s = (1 if li == 0 else 2) if bi == 0 else 1
        def step(inp, inpo, inpe, conv, bn, k, st_, pd, relu=True):
            y, stt = dev_conv(inp, conv, k, st_, pd)
            yo = F.conv2d(inpo, params[conv + ".weight"], None, st_, pd)
            ye = F.conv2d(inpe, emu(params[conv + ".weight"]), None, st_, pd)
            return y, stt, yo, ye
        y1, s1, y1o, y1e = step(a, ao, ae, p + ".conv1", p + ".bn1", 1, 1, 0)
        C1 = y1.shape[-1]
        a1 = ops.bn_act(y1, params[p+".bn1.weight"].cuda(), params[p+".bn1.bias"].cuda(), stats=s1, count=y1.numel()//C1)
        a1o = obn(y1o, p+".bn1"); a1e = emu(F.relu(ebn(y1e, p+".bn1")))
        y2, s2, y2o, y2e = step(a1, a1o, a1e, p + ".conv2", p + ".bn2", 3, s, 1)
        a2 = ops.bn_act(y2, params[p+".bn2.weight"].cuda(), params[p+".bn2.bias"].cuda(), stats=s2, count=y2.numel()//C1)
        a2o = obn(y2o, p+".bn2"); a2e = emu(F.relu(ebn(y2e, p+".bn2")))
        y3, s3, y3o, y3e = step(a2, a2o, a2e, p + ".conv3", p + ".bn3", 1, 1, 0)
        C3 = y3.shape[-1]; cnt = y3.numel()//C3
        if p + ".downsample.0.weight" in params:
            yd, sd, ydo, yde = step(a, ao, ae, p + ".downsample.0", p + ".downsample.1", 1, s, 0)
            out = ops.bn_act(y3, params[p+".bn3.weight"].cuda(), params[p+".bn3.bias"].cuda(), stats=s3, count=cnt, res=yd,
                             res_bn=dict(gamma=params[p+".downsample.1.weight"].cuda(), beta=params[p+".downsample.1.bias"].cuda(), stats=sd))
            outo = F.relu(obn(y3o, p+".bn3", False) + obn(ydo, p+".downsample.1", False))
            oute = emu(F.relu(ebn(y3e, p+".bn3") + ebn(yde, p+".downsample.1")))
        else:
            out = ops.bn_act(y3, params[p+".bn3.weight"].cuda(), params[p+".bn3.bias"].cuda(), stats=s3, count=cnt, res=a)
            outo = F.relu(obn(y3o, p+".bn3", False) + ao)
            oute = emu(F.relu(ebn(y3e, p+".bn3") + ae))
        print(p, "vs fp32 (max,rms)", "%.4f %.4f" % rel(out.permute(0,3,1,2), outo), " vs bf16-emu", "%.4f %.4f" % rel(out.permute(0,3,1,2), oute),
              " emu vs fp32 %.4f %.4f" % rel(oute, outo))
        a, ao, ae = out, outo, oute
