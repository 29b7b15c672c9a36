"""Encoder: drop-in for the reference's ``cnn.ResNet`` (cnn.py:9-51) and
``Attention/cnn_attn.ResNet`` (cnn_attn.py:9-52) on MI355X.

Same constructor arguments, same public attributes (``model``,
``linear_secondlast_layer``, ``last_layer`` -- read by name at main.py:96) and the
same ``state_dict`` keys as the reference, so reference
checkpoints load.  The arithmetic does not run in torch: ``forward`` issues one
C-ABI call (``st_resnet_forward``) that drives the hand-written HIP kernels, and
the trainable head (Linear -> BatchNorm1d) runs in ``st_linear_bn1d_*``.

The parameter-holding sub-modules are ordinary ``nn.Conv2d`` / ``nn.BatchNorm2d``
objects used as *containers only* (names, shapes, inits); they are never called.
BatchNorm parameters and buffers are views into flat fp32 arrays laid out in the
engine's layer order, so the whole backbone is described to the C ABI by five
pointers.

Offline note: the reference downloads ImageNet weights (``pretrained=True``,
cnn.py:23-31).  There is no network here; weights are random (torchvision's
Kaiming fan-out init) until ``load_state_dict`` supplies real ones.
"""
import ctypes as C
import math

import torch
import torch.nn as nn

from . import _lib, ops
from ._lib import ST_BF16, ST_F32, check, lib

_SPECS = {18: ("basic", [2, 2, 2, 2]), 34: ("basic", [3, 4, 6, 3]), 50: ("bottleneck", [3, 4, 6, 3]),
          101: ("bottleneck", [3, 4, 23, 3]), 152: ("bottleneck", [3, 8, 36, 3])}


class _Block(nn.Module):
    """Parameter container with torchvision's Bottleneck / BasicBlock attribute names."""

    def __init__(self, kind, inpl, planes, stride, downsample):
        super().__init__()
        if kind == "bottleneck":
            self.conv1 = nn.Conv2d(inpl, planes, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
            self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
            self.bn3 = nn.BatchNorm2d(planes * 4)
        else:
            self.conv1 = nn.Conv2d(inpl, planes, 3, stride, 1, bias=False)
            self.bn1 = nn.BatchNorm2d(planes)
            self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
            self.bn2 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def pairs(self):
        out = [(self.conv1, self.bn1), (self.conv2, self.bn2)]
        if hasattr(self, "conv3"):
            out.append((self.conv3, self.bn3))
        if self.downsample is not None:
            out.append((self.downsample[0], self.downsample[1]))
        return out

    def forward(self, x):  # pragma: no cover - containers are never called
        raise _lib.ShowTellHipError("backbone sub-modules are parameter containers; call ResNet.forward")


def _build_backbone(version, avgpool):
    if version not in _SPECS:
        raise ValueError("Please specify a valid ResNet version. %d doesn't exist." % (version))  # cnn.py:33
    kind, nblocks = _SPECS[version]
    exp = 4 if kind == "bottleneck" else 1
    children = [nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64), nn.ReLU(inplace=True), nn.MaxPool2d(3, 2, 1)]
    inpl = 64
    for li, (planes, nb) in enumerate(zip([64, 128, 256, 512], nblocks)):
        blocks = []
        for bi in range(nb):
            s = (1 if li == 0 else 2) if bi == 0 else 1
            ds = None
            if bi == 0 and (s != 1 or inpl != planes * exp):
                ds = nn.Sequential(nn.Conv2d(inpl, planes * exp, 1, s, bias=False), nn.BatchNorm2d(planes * exp))
            blocks.append(_Block(kind, inpl, planes, s, ds))
            inpl = planes * exp
        children.append(nn.Sequential(*blocks))
    if avgpool:
        children.append(nn.AdaptiveAvgPool2d((1, 1)))
    model = nn.Sequential(*children)
    for m in model.modules():  # torchvision resnet init
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
    return model, inpl


class _Backbone:
    """Engine-side state of one ResNet: C handle, packed weights, flat BN arrays, workspace."""

    def __init__(self, model, version, dtype):
        self.version, self.dtype = version, dtype
        self.pairs = [(model[0], model[1])]
        for li in range(4, 8):
            for blk in model[li]:
                self.pairs += blk.pairs()
        self.handle = None
        self.flat = None          # dict of flat fp32 BN arrays
        self.packed = None
        self.packed_key = None
        self.ws = {}
        self._upd_event = None

    def _ensure_handle(self):
        if self.handle is None:
            h = C.c_void_p()
            rc = lib().st_resnet_create(self.version, ST_BF16 if self.dtype == torch.bfloat16 else ST_F32, C.byref(h))
            if rc == 2:
                raise ValueError(lib().st_last_error().decode())
            check(rc, "st_resnet_create")
            self.handle = h
            n = lib().st_resnet_num_convs(h)
            assert n == len(self.pairs), (n, len(self.pairs))
            self.info = []
            for i in range(n):
                v = [C.c_int() for _ in range(6)]
                o = [C.c_size_t(), C.c_size_t()]
                ko, fo, fn = C.c_int(), C.c_size_t(), C.c_int()
                check(lib().st_resnet_conv_info(h, i, *[C.byref(a) for a in v], *[C.byref(a) for a in o], C.byref(ko), C.byref(fo), C.byref(fn)), "conv_info")
                self.info.append(dict(cin=v[0].value, cout=v[1].value, k=v[2].value, stride=v[3].value, pad=v[4].value,
                                      cin_p=v[5].value, woff=o[0].value, bnoff=o[1].value, korder=ko.value,
                                      woff_frag=fo.value, ntw=fn.value))
                conv = self.pairs[i][0]
                assert tuple(conv.weight.shape) == (v[1].value, v[0].value, v[2].value, v[2].value)
        return self.handle

    def flatten_bn(self, device):
        """(Re)build the flat BN arrays on `device` and point every BN tensor at its view."""
        self._ensure_handle()
        total = lib().st_resnet_bn_channels(self.handle)
        new = {k: torch.empty(total, device=device, dtype=torch.float32) for k in ("gamma", "beta", "rm", "rv")}
        nbt = torch.zeros(len(self.pairs), device=device, dtype=torch.long)
        for i, (_, bn) in enumerate(self.pairs):
            o, c = self.info[i]["bnoff"], self.info[i]["cout"]
            for key, t in (("gamma", bn.weight), ("beta", bn.bias), ("rm", bn.running_mean), ("rv", bn.running_var)):
                view = new[key][o:o + c]
                view.copy_(t.data)
                t.data = view
            nbt[i] = bn.num_batches_tracked.item() if bn.num_batches_tracked.numel() else 0
            bn.num_batches_tracked.data = nbt[i]
        self.flat, self.nbt = new, nbt

    def _bn_ok(self, device):
        if self.flat is None or self.flat["gamma"].device != device:
            return False
        bn = self.pairs[-1][1]
        o = self.info[-1]["bnoff"]
        return bn.weight.data_ptr() == self.flat["gamma"][o:].data_ptr()

    def pack_weights(self, device):
        key = (device, tuple(c.weight._version for c, _ in self.pairs), tuple(c.weight.data_ptr() for c, _ in self.pairs))
        if self.packed is not None and key == self.packed_key:
            return
        self._ensure_handle()
        n = lib().st_resnet_weight_elems(self.handle)
        self.packed = torch.empty(n, device=device, dtype=self.dtype)
        es = self.packed.element_size()
        dt = ST_BF16 if self.dtype == torch.bfloat16 else ST_F32
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        for i, (conv, _) in enumerate(self.pairs):
            inf = self.info[i]
            w = conv.weight.data
            assert w.is_cuda and w.dtype == torch.float32 and w.is_contiguous()
            check(lib().st_pack_conv_weight(C.c_void_p(w.data_ptr()), C.c_void_p(self.packed.data_ptr() + inf["woff"] * es),
                                            dt, inf["cout"], inf["cin"], inf["k"], inf["k"], inf["cin_p"], inf["korder"], st),
                  "st_pack_conv_weight")
            if inf["ntw"] > 0:      # second, fragment-major copy for the image-resident 3x3 kernel (st_conv3x3_img)
                check(lib().st_pack_conv_weight_frag(C.c_void_p(w.data_ptr()), C.c_void_p(self.packed.data_ptr() + inf["woff_frag"] * es),
                                                     inf["cout"], inf["cin"], inf["k"], inf["k"], inf["ntw"], st), "st_pack_conv_weight_frag")
        self.packed_key = key

    def block_outputs(self, x, train):
        """Diagnosis / test aid (st_resnet_set_taps): one forward that also returns every residual block's output as the engine
        stored it -- a list of (B, h, w, C) NHWC tensors in the compute dtype, in network order."""
        self._ensure_handle()
        B, _, H, W = x.shape
        h = ((H + 6 - 7) // 2 + 1 - 1) // 2 + 1
        w = ((W + 6 - 7) // 2 + 1 - 1) // 2 + 1
        kind, nblocks = _SPECS[self.version]
        exp = 4 if kind == "bottleneck" else 1
        shapes = []
        for li, (planes, nb) in enumerate(zip([64, 128, 256, 512], nblocks)):
            for bi in range(nb):
                if li > 0 and bi == 0:
                    h, w = (h - 1) // 2 + 1, (w - 1) // 2 + 1
                shapes.append((B, h, w, planes * exp))
        n = sum(a * b * c * d for a, b, c, d in shapes)
        buf = torch.empty(n, device=x.device, dtype=self.dtype)
        check(lib().st_resnet_set_taps(self.handle, C.c_void_p(buf.data_ptr()), buf.numel() * buf.element_size()), "st_resnet_set_taps")
        try:
            pooled, _ = self.forward(x, train, True, False)
            torch.cuda.synchronize()
        finally:
            check(lib().st_resnet_set_taps(self.handle, None, 0), "st_resnet_set_taps")
        outs, o = [], 0
        for sh in shapes:
            k = sh[0] * sh[1] * sh[2] * sh[3]
            outs.append(buf[o:o + k].view(sh))
            o += k
        return outs, pooled

    def forward(self, x, train, want_pooled, want_ncp, pooled_dtype=torch.float32):
        if not x.is_cuda:
            raise _lib.ShowTellHipError("ResNet.forward needs a HIP device tensor (no CPU fallback in the MI355X build)")
        if x.dim() != 4 or x.shape[1] != 3:
            raise _lib.ShowTellHipError(f"expected (B,3,H,W) images, got {tuple(x.shape)}")
        x = x.detach().contiguous().float()
        dev = x.device
        if not self._bn_ok(dev):
            self.flatten_bn(dev)
        self.pack_weights(dev)
        B, _, H, W = x.shape
        # one workspace per (shape, stream): forwards of successive minibatches may run concurrently on different streams
        # (train.py pipelines two of them); at most four are kept
        cur = torch.cuda.current_stream()
        key = (dev, B, H, W, cur.cuda_stream)
        if key not in self.ws:
            nbytes = lib().st_resnet_workspace_bytes(self.handle, B, H, W)
            if nbytes == 0:
                raise _lib.ShowTellHipError(f"unsupported input size {tuple(x.shape)}")
            if any(k[:4] != key[:4] for k in self.ws):   # another input shape: its workspaces go (after their streams drained)
                torch.cuda.synchronize(dev)
                self.ws = {k: v for k, v in self.ws.items() if k[:4] == key[:4]}
            if len(self.ws) >= 4:
                torch.cuda.synchronize(dev)             # rare: the evicted workspace may still be in use on its stream
                self.ws.pop(next(iter(self.ws)))
            with torch.cuda.stream(cur):
                self.ws[key] = torch.empty(nbytes, device=dev, dtype=torch.uint8)
        ws = self.ws[key]
        F = lib().st_resnet_feat_dim(self.handle)
        ho = ((H + 6 - 7) // 2 + 1 - 1) // 2 + 1
        wo = ((W + 6 - 7) // 2 + 1 - 1) // 2 + 1
        for _ in range(3):
            ho, wo = (ho - 1) // 2 + 1, (wo - 1) // 2 + 1
        pooled = torch.empty(B, F, device=dev, dtype=pooled_dtype) if want_pooled else None
        ncp = torch.empty(B, F, ho * wo, device=dev, dtype=torch.float32) if want_ncp else None
        p = lambda t: None if t is None else C.c_void_p(t.data_ptr())
        fl = self.flat
        check(lib().st_resnet_forward(self.handle, p(x), B, H, W, p(self.packed), p(fl["gamma"]), p(fl["beta"]),
                                      p(fl["rm"]), p(fl["rv"]), 2 if train else 0, 0.1, 1e-5, p(ws), ws.numel(),
                                      None, p(pooled), ST_BF16 if pooled_dtype == torch.bfloat16 else ST_F32, p(ncp),
                                      C.c_void_p(cur.cuda_stream)), "st_resnet_forward")
        if train:
            # the momentum updates of the running buffers (and num_batches_tracked) are applied in minibatch order even when
            # the forwards themselves overlap on different streams: an event chain over the small update kernel
            if self._upd_event is not None:
                cur.wait_event(self._upd_event)
            check(lib().st_resnet_update_running(self.handle, p(ws), p(fl["rm"]), p(fl["rv"]), 0.1, C.c_void_p(cur.cuda_stream)),
                  "st_resnet_update_running")
            self.nbt.add_(1)
            self._upd_event = torch.cuda.Event()
            self._upd_event.record(cur)
        return pooled, ncp

    def __del__(self):
        try:
            if self.handle is not None:
                lib().st_resnet_destroy(self.handle)
        except Exception:
            pass


class ResNet(nn.Module):
    '''
    Encoding via ResNet (reference cnn.py:9-51); `dtype` selects the kernels' storage type
    (torch.float32 = parity mode, torch.bfloat16 = performance mode, fp32 accumulation).
    '''

    _AVGPOOL = True

    def __init__(self, resnet_version=101, embed_dim=256, dtype=torch.float32):
        super(ResNet, self).__init__()
        self.model, feat = _build_backbone(resnet_version, self._AVGPOOL)
        # Training only the last 2 layers, i.e. linear and batchnorm layer (cnn.py:36-38)
        self.linear_secondlast_layer = nn.Linear(feat, embed_dim)
        self.last_layer = nn.BatchNorm1d(embed_dim, momentum=0.01)
        self.linear_secondlast_layer.weight.data.normal_(0, 0.05)   # cnn.py:41
        self.last_layer.bias.data.fill_(0)                           # cnn.py:42
        self.compute_dtype = dtype
        self._bb = _Backbone(self.model, resnet_version, dtype)

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self._bb.flat = None      # .cuda()/.cpu()/.to() replaced the tensors: re-flatten lazily
        self._bb.packed = None
        return out

    def backbone_features(self, x):
        """Pooled (B,F) fp32 backbone output, detached (cnn.py:46-48)."""
        pooled, _ = self._bb.forward(x, self.training, True, False)
        return pooled

    def forward(self, x):
        from .head import linear_bn1d
        f = self.backbone_features(x)   # x = Variable(x.data): no gradient into the backbone (cnn.py:47)
        return linear_bn1d(f, self.linear_secondlast_layer, self.last_layer, self.training, self.compute_dtype)
