"""The stem at B = 128, 224 x 224: two-kernel form (st_conv on the blocked image + st_maxpool3x3s2_bn) against st_stem_conv_pool
(debug aid; bench.py is the contract)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from showtell_amd import ops, _lib
import ctypes as C

B, H, W = 128, 224, 224
dt = torch.bfloat16
x = torch.randn(B, 3, H, W, device="cuda")
w = torch.randn(64, 8, 7, 7, device="cuda") / 147 ** 0.5
w[:, 3:] = 0
wd = ops.pack_conv_weight(w, dt)
gamma = torch.ones(64, device="cuda"); gamma[::3] = -1
beta = torch.zeros(64, device="cuda")
lib = _lib.lib()
xs = torch.empty(B, H // 2 + 3, W // 2 + 3, 16, device="cuda", dtype=dt)
_lib.check(lib.st_nchw_to_s2d16(x.data_ptr(), xs.data_ptr(), 1 if dt == torch.bfloat16 else 0, B, H, W, None), "s2d")
ws = torch.empty(64, 256, device="cuda", dtype=dt)
_lib.check(lib.st_stem_weight_s2d(wd.data_ptr(), ws.data_ptr(), ops._DT[dt], 8, None), "w")
wf = torch.empty(64 * 256, device="cuda", dtype=dt)
_lib.check(lib.st_stem_weight_frag(ws.data_ptr(), wf.data_ptr(), None), "wf")
raw = torch.empty(B, H // 2, W // 2, 64, device="cuda", dtype=dt)
pooled = torch.empty(B, 56, 56, 64, device="cuda", dtype=dt)
s0 = torch.zeros(64, 128, device="cuda")
n = float(B * 112 * 112)


def two():
    d = _lib.ConvDesc(xs.data_ptr(), ws.data_ptr(), raw.data_ptr(), None, None, None, None, s0.data_ptr(),
                      ops._DT[dt], ops._DT[dt], B, H // 2 + 3, W // 2 + 3, 64, H // 2, W // 2, 64, 4, 1, 1, 0, 16, 256, 64, 0, 0, 36, 0, 64)
    _lib.check(lib.st_conv(C.byref(d), None), "conv")
    _lib.check(lib.st_maxpool3x3s2_bn(raw.data_ptr(), pooled.data_ptr(), ops._DT[dt], B, 112, 112, 64, s0.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                      None, None, n, 1e-5, None), "pool")


def one():
    d = _lib.StemConvPoolDesc(xs.data_ptr(), wf.data_ptr(), pooled.data_ptr(), s0.data_ptr(), 8, gamma.data_ptr(), None, None, B, H, W)
    _lib.check(lib.st_stem_conv_pool(C.byref(d), None), "stem")


for f in (two, one):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record(); torch.cuda.synchronize()
    print(f"{f.__name__}: {e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
