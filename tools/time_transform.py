"""Times the device-side input transform (show-tell_amd/data.py) on a COCO-shaped minibatch of 128 uint8 images:
whole call from host arrays (staging + PCIe + 2 kernels) and the two kernels alone (HIP events, inputs resident)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from showtell_amd import _lib
from showtell_amd.data import DeviceTransform

B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
shapes = [(480, 640), (640, 480), (427, 640), (375, 500)]
rng = np.random.default_rng(0)
imgs = [rng.integers(0, 256, size=shapes[i % 4] + (3,), dtype=np.uint8) for i in range(B)]
tf = DeviceTransform()
for _ in range(3):
    out = tf(imgs)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 10
for _ in range(n):
    out = tf(imgs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / n
src_bytes = sum(a.size for a in imgs)
print(f"host->features-ready: {dt*1e3:.2f} ms per {B} images ({B/dt:.0f} images/s), {src_bytes/1e6:.1f} MB uint8 over PCIe "
      f"(the reference's float32 tensors: {B*3*224*224*4/1e6:.1f} MB)")

# kernels alone
dev = torch.device("cuda")
src = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs])).to(dev)
sizes = np.asarray([a.size for a in imgs], np.int64)
off = torch.from_numpy(np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)).to(dev)
meta = torch.tensor([[a.shape[0] for a in imgs], [a.shape[1] for a in imgs], [i % 4 for i in range(B)]], dtype=torch.int32, device=dev)
lut = tf._lut
mh, mw = int(meta[0].max()), int(meta[1].max())
tmp = torch.empty(B * mh * 224 * 3, dtype=torch.uint8, device=dev)
o = torch.empty(B, 3, 224, 224, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
d = _lib.ImageBatchDesc(p(src), src.numel(), p(off), p(meta[0]), p(meta[1]), p(meta[2]), B, mh, mw, 224, 224, p(lut), p(tmp), p(o), None)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(3):
    _lib.check(_lib.lib().st_image_transform(C.byref(d), st), "st_image_transform")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    _lib.lib().st_image_transform(C.byref(d), st)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
inter = sum(a.shape[0] * 224 * 3 for a in imgs)
alg = src_bytes + 2 * inter + B * 3 * 224 * 224 * 4
print(f"kernels: {us:.1f} us per {B} images; algorithmic bytes {alg/1e6:.1f} MB -> {alg/us/1e3:.0f} GB/s ({alg/us/1e3/8000*100:.1f} % of 8 TB/s)")
