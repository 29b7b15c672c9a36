"""Tensor-level wrappers over the C ABI (torch supplies device memory and streams only).

Every function checks shapes/dtypes on the host before a kernel is launched and
raises ``ShowTellHipError`` on failure.  All tensors must live on a HIP device.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import (ST_BF16, ST_F32, BnActDesc, ConvB2bDesc, Conv1x1KfuseDesc, Conv1x1WregDesc, Conv3x3ImgDesc, ConvDesc, StemConvPoolDesc,
                   check, lib)

_DT = {torch.float32: ST_F32, torch.bfloat16: ST_BF16}


def dt_code(t):
    if t.dtype not in _DT:
        raise _lib.ShowTellHipError(f"unsupported dtype {t.dtype}")
    return _DT[t.dtype]


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _lib.ShowTellHipError("tensor is not on the GPU (the HIP path has no CPU fallback)")
        if t is not None and not t.is_contiguous():
            raise _lib.ShowTellHipError("tensor must be contiguous")


def conv_nhwc(x, w, KH, KW, stride, pad, out_dtype=None, bias=None, scale=None, shift=None,
              residual=None, stats=None, relu=False, out=None, accumulate=False, k_order=0, stats_replicas=0,
              in_bn=None):
    """x: (B,Hin,Win,Cin) NHWC; w: (N, KH*KW*Cin) K-contiguous.  Returns (B,Ho,Wo,N).
    stats_replicas = R > 1: stats is (R, 2N), pixel tile t adds into replica t % R.
    in_bn = dict(stats, gamma, beta, count[, eps]): x is a producer's raw output, the conv reads relu(batchnorm(x))."""
    _dev(x, w, bias, scale, shift, residual, stats, out)
    B, Hin, Win, Cin = x.shape
    N = w.shape[0]
    assert w.shape[1] == KH * KW * Cin, (w.shape, KH, KW, Cin)
    Ho = (Hin + 2 * pad - KH) // stride + 1
    Wo = (Win + 2 * pad - KW) // stride + 1
    out_dtype = out_dtype or x.dtype
    if out is None:
        out = torch.empty(B, Ho, Wo, N, device=x.device, dtype=out_dtype)
    assert out.shape == (B, Ho, Wo, N) and out.dtype == out_dtype
    d = ConvDesc(_p(x), _p(w), _p(out), _p(bias), _p(scale), _p(shift), _p(residual), _p(stats),
                 dt_code(x), _DT[out_dtype], B, Hin, Win, Cin, Ho, Wo, N, KH, KW, stride, pad,
                 Cin, w.shape[1], N, int(relu), int(accumulate), 0, int(k_order), int(stats_replicas))
    if in_bn is not None:
        _dev(in_bn["stats"], in_bn["gamma"], in_bn["beta"])
        d.in_stats, d.in_gamma, d.in_beta = in_bn["stats"].data_ptr(), in_bn["gamma"].data_ptr(), in_bn["beta"].data_ptr()
        d.in_count, d.in_eps = float(in_bn["count"]), float(in_bn.get("eps", 1e-5))
    check(lib().st_conv(C.byref(d), _stream()), "st_conv")
    return out


def stem_conv_s2d(images, w_packed, cpad, dtype, stats=None, scale=None, shift=None, relu=False):
    """torchvision stem (7x7 s2 p3 over RGB, cnn.py:46) through the space-to-depth route: images (B,3,H,W) fp32 with
    even H, W; w_packed the usual (64, 7*7*cpad) k_order-0 weights.  Returns (B,H/2,W/2,64)."""
    _dev(images, w_packed, stats, scale, shift)
    B, Cc, H, W = images.shape
    assert Cc == 3 and images.dtype == torch.float32
    xs = torch.empty(B, H // 2 + 3, W // 2 + 3, 16, device=images.device, dtype=dtype)
    check(lib().st_nchw_to_s2d16(_p(images), _p(xs), _DT[dtype], B, H, W, _stream()), "st_nchw_to_s2d16")
    ws = torch.empty(64, 256, device=images.device, dtype=dtype)
    check(lib().st_stem_weight_s2d(_p(w_packed), _p(ws), _DT[dtype], cpad, _stream()), "st_stem_weight_s2d")
    out = torch.empty(B, H // 2, W // 2, 64, device=images.device, dtype=dtype)
    d = ConvDesc(_p(xs), _p(ws), _p(out), None, _p(scale), _p(shift), None, _p(stats),
                 _DT[dtype], _DT[dtype], B, H // 2 + 3, W // 2 + 3, 64, H // 2, W // 2, 64, 4, 1, 1, 0,
                 16, 256, 64, int(relu), 0, 36, 0, 0)
    check(lib().st_conv(C.byref(d), _stream()), "st_conv(s2d stem)")
    return out, xs, ws


def stem_conv_pool(images, w_packed, cpad, stats=None, stats_replicas=0, gamma=None, scale=None, shift=None):
    """The bf16 stem in one kernel (st_stem_conv_pool): conv 7x7/2 + statistics + maxpool 3x3/2.  images (B,3,H,W) fp32, even H, W;
    w_packed the usual (64, 7*7*cpad) bf16 k_order-0 weights.  Train (gamma, stats): the pooled RAW output (max / min by sign(gamma));
    eval (scale, shift): maxpool(relu(conv * scale + shift)).  Returns (B, PH, PW, 64) bf16."""
    _dev(images, w_packed, stats, gamma, scale, shift)
    B, Cc, H, W = images.shape
    assert Cc == 3 and images.dtype == torch.float32
    dt = torch.bfloat16
    xs = torch.empty(B, H // 2 + 3, W // 2 + 3, 16, device=images.device, dtype=dt)
    check(lib().st_nchw_to_s2d16(_p(images), _p(xs), _DT[dt], B, H, W, _stream()), "st_nchw_to_s2d16")
    ws = torch.empty(64, 256, device=images.device, dtype=dt)
    check(lib().st_stem_weight_s2d(_p(w_packed), _p(ws), _DT[dt], cpad, _stream()), "st_stem_weight_s2d")
    wf = torch.empty(64 * 256, device=images.device, dtype=dt)
    check(lib().st_stem_weight_frag(_p(ws), _p(wf), _stream()), "st_stem_weight_frag")
    wf1 = torch.empty_like(wf)                     # the one-launch form st_resnet_forward uses: must be the same operands
    check(lib().st_stem_weight_frag_packed(_p(w_packed), int(cpad), _p(wf1), _stream()), "st_stem_weight_frag_packed")
    if not torch.equal(wf, wf1):
        raise _lib.ShowTellHipError("st_stem_weight_frag_packed differs from st_stem_weight_frag(st_stem_weight_s2d(.))")
    PH, PW = (H // 2 - 1) // 2 + 1, (W // 2 - 1) // 2 + 1
    out = torch.empty(B, PH, PW, 64, device=images.device, dtype=dt)
    d = StemConvPoolDesc(_p(xs), _p(wf), _p(out), _p(stats), int(stats_replicas), _p(gamma), _p(scale), _p(shift), B, H, W)
    check(lib().st_stem_conv_pool(C.byref(d), _stream()), "st_stem_conv_pool")
    return out


def gemm_nt(a, w, out_dtype=None, bias=None, out=None, accumulate=False, stats=None, relu=False,
            lda=None, ldw=None, K=None, split_k=0):
    """y[M,N] = a[M,K] @ w[N,K]^T (+bias).  a, w may carry padded leading dimensions."""
    _dev(a, w, bias, out, stats)
    M = a.shape[0]
    N = w.shape[0]
    lda = lda or a.stride(0)
    ldw = ldw or w.stride(0)
    K = K or a.shape[1]
    out_dtype = out_dtype or a.dtype
    if out is None:
        out = torch.empty(M, N, device=a.device, dtype=out_dtype)
    ldy = out.stride(0)
    d = ConvDesc(_p(a), _p(w), _p(out), _p(bias), None, None, None, _p(stats),
                 dt_code(a), _DT[out.dtype], M, 1, 1, K, 1, 1, N, 1, 1, 1, 0,
                 lda, ldw, ldy, int(relu), int(accumulate), 0, 0)
    d.split_k = int(split_k)
    check(lib().st_conv(C.byref(d), _stream()), "st_conv(gemm)")
    return out


def gemm_nt_batch(As, Ws, outs, accumulate=False):
    """Several y_i[M,N] (+)= a_i[M,K] @ w_i[N,K]^T of identical shape as ONE launch (st_conv_batch)."""
    n = len(As)
    arr = (ConvDesc * n)()
    for i, (a, w, o) in enumerate(zip(As, Ws, outs)):
        _dev(a, w, o)
        M, K = a.shape
        N = w.shape[0]
        arr[i] = ConvDesc(_p(a), _p(w), _p(o), None, None, None, None, None, dt_code(a), _DT[o.dtype], M, 1, 1, K, 1, 1, N, 1, 1, 1, 0,
                          a.stride(0), w.stride(0), o.stride(0), 0, int(accumulate), 0, 0, 0)
    check(lib().st_conv_batch(arr, n, _stream()), "st_conv_batch")
    return outs


def bn_act(x, gamma, beta, stats=None, running=None, count=1.0, eps=1e-5, relu=True, res=None,
           res_bn=None, out=None, stats_replicas=0):
    """x: (..., C) channels-last.  res_bn = dict(gamma, beta, stats | running=(mean,var)) or None."""
    _dev(x, gamma, beta, stats, res, out)
    Cc = x.shape[-1]
    rows = x.numel() // Cc
    if out is None:
        out = torch.empty_like(x)
    rm, rv = running if running is not None else (None, None)
    rb = res_bn or {}
    rrm, rrv = rb.get("running", (None, None))
    d = BnActDesc(_p(x), _p(out), _p(res), _p(stats), _p(gamma), _p(beta), _p(rm), _p(rv),
                  _p(rb.get("stats")), _p(rb.get("gamma")), _p(rb.get("beta")), _p(rrm), _p(rrv),
                  int(res_bn is not None), dt_code(x), rows, Cc, float(count), float(eps), int(relu),
                  int(stats_replicas), int(rb.get("stats_replicas", 0)))
    check(lib().st_bn_act(C.byref(d), _stream()), "st_bn_act")
    return out


def bn_update_running(stats, running_mean, running_var, count, momentum):
    _dev(stats, running_mean, running_var)
    check(lib().st_bn_update_running(_p(stats), _p(running_mean), _p(running_var), running_mean.numel(),
                                     float(count), float(momentum), _stream()), "st_bn_update_running")


def nchw_to_nhwc(x, dtype, cpad):
    _dev(x)
    assert x.dtype == torch.float32
    B, Cc, H, W = x.shape
    y = torch.empty(B, H, W, cpad, device=x.device, dtype=dtype)
    check(lib().st_nchw_to_nhwc(_p(x), _p(y), _DT[dtype], B, Cc, H, W, cpad, _stream()), "st_nchw_to_nhwc")
    return y


def nhwc_to_ncp_f32(x):
    _dev(x)
    B, H, W, Cc = x.shape
    y = torch.empty(B, Cc, H * W, device=x.device, dtype=torch.float32)
    check(lib().st_nhwc_to_ncp_f32(_p(x), _p(y), dt_code(x), B, H * W, Cc, _stream()), "st_nhwc_to_ncp_f32")
    return y


def maxpool3x3s2(x):
    _dev(x)
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=x.dtype)
    check(lib().st_maxpool3x3s2(_p(x), _p(y), dt_code(x), B, H, W, Cc, _stream()), "st_maxpool3x3s2")
    return y


def maxpool3x3s2_bn(x, gamma, beta, stats=None, running=None, count=1.0, eps=1e-5):
    """maxpool3x3s2(relu(batchnorm(x))) in one pass (the ResNet stem tail)."""
    _dev(x, gamma, beta, stats)
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    y = torch.empty(B, Ho, Wo, Cc, device=x.device, dtype=x.dtype)
    rm, rv = running if running is not None else (None, None)
    check(lib().st_maxpool3x3s2_bn(_p(x), _p(y), dt_code(x), B, H, W, Cc, _p(stats), _p(gamma), _p(beta), _p(rm), _p(rv),
                                   float(count), float(eps), _stream()), "st_maxpool3x3s2_bn")
    return y


def global_avgpool(x, out_dtype=None):
    _dev(x)
    B, H, W, Cc = x.shape
    out_dtype = out_dtype or x.dtype
    y = torch.empty(B, Cc, device=x.device, dtype=out_dtype)
    check(lib().st_global_avgpool(_p(x), _p(y), dt_code(x), _DT[out_dtype], B, H * W, Cc, _stream()), "st_global_avgpool")
    return y


def cast(x, dtype, out=None):
    _dev(x, out)
    if out is None:
        out = torch.empty(x.shape, device=x.device, dtype=dtype)
    check(lib().st_cast(_p(x), _p(out), dt_code(x), _DT[dtype], x.numel(), _stream()), "st_cast")
    return out


def transpose(x, ldy=None, out=None, colsum=None):
    """x: (rows, cols) -> (cols, ldy) with ldy >= rows, zero padded; colsum (cols,) fp32 += column sums of x."""
    _dev(x, out, colsum)
    rows, cols = x.shape
    ldy = ldy or rows
    if out is None:
        out = torch.empty(cols, ldy, device=x.device, dtype=x.dtype)
    check(lib().st_transpose_colsum(_p(x), _p(out), _p(colsum), dt_code(x), rows, cols, x.stride(0), ldy, _stream()),
          "st_transpose_colsum")
    return out


def pack_conv_weight_frag(w, ntw):
    """(Cout,Cin,KH,KW) fp32 torch layout -> fragment-major bf16 (st_pack_conv_weight_frag) for the image-resident kernels."""
    _dev(w)
    Cout, Cin, KH, KW = w.shape
    out = torch.empty(Cout * KH * KW * Cin, device=w.device, dtype=torch.bfloat16)
    check(lib().st_pack_conv_weight_frag(_p(w), _p(out), Cout, Cin, KH, KW, int(ntw), _stream()), "st_pack_conv_weight_frag")
    return out


def conv3x3_img_supported(H, W, C, N):
    return int(lib().st_conv3x3_img_supported(H, W, C, N))


def conv3x3_img(x, w_frag, N, stats=None, stats_replicas=0, scale=None, shift=None, relu=False, in_bn=None, out=None):
    """Image-resident 3x3 s1 p1 conv (st_conv3x3_img): x (B,H,W,C) bf16 NHWC, w_frag from pack_conv_weight_frag."""
    _dev(x, w_frag, stats, scale, shift, out)
    B, H, W, Cc = x.shape
    if x.dtype != torch.bfloat16:
        raise _lib.ShowTellHipError("conv3x3_img is a bf16 kernel")
    if out is None:
        out = torch.empty(B, H, W, N, device=x.device, dtype=torch.bfloat16)
    d = Conv3x3ImgDesc(_p(x), _p(w_frag), _p(out), _p(stats), int(stats_replicas), _p(scale), _p(shift), int(relu),
                       None, None, None, 0.0, 0.0, B, H, W, Cc, N, 0)
    if in_bn is not None:
        _dev(in_bn["stats"], in_bn["gamma"], in_bn["beta"])
        d.in_stats, d.in_gamma, d.in_beta = in_bn["stats"].data_ptr(), in_bn["gamma"].data_ptr(), in_bn["beta"].data_ptr()
        d.in_count, d.in_eps = float(in_bn["count"]), float(in_bn.get("eps", 1e-5))
        d.in_stats_replicas = int(in_bn.get("replicas", 0))
    check(lib().st_conv3x3_img(C.byref(d), _stream()), "st_conv3x3_img")
    return out


def conv3x3_s2_supported(Cin, N):
    return int(lib().st_conv3x3_s2_supported(Cin, N))


def conv3x3_s2(x, w_frag, N, stats=None, stats_replicas=0, scale=None, shift=None, relu=False, in_bn=None, out=None):
    """K-streaming 3x3 stride-2 pad-1 conv (st_conv3x3_s2): x (B,H,W,C) bf16 NHWC, w_frag = pack_conv_weight_frag(w, conv3x3_s2_supported(C, N))."""
    _dev(x, w_frag, stats, scale, shift, out)
    B, H, W, Cc = x.shape
    if x.dtype != torch.bfloat16:
        raise _lib.ShowTellHipError("conv3x3_s2 is a bf16 kernel")
    if out is None:
        out = torch.empty(B, (H - 1) // 2 + 1, (W - 1) // 2 + 1, N, device=x.device, dtype=torch.bfloat16)
    d = Conv3x3ImgDesc(_p(x), _p(w_frag), _p(out), _p(stats), int(stats_replicas), _p(scale), _p(shift), int(relu),
                       None, None, None, 0.0, 0.0, B, H, W, Cc, N, 0)
    if in_bn is not None:
        _dev(in_bn["stats"], in_bn["gamma"], in_bn["beta"])
        d.in_stats, d.in_gamma, d.in_beta = in_bn["stats"].data_ptr(), in_bn["gamma"].data_ptr(), in_bn["beta"].data_ptr()
        d.in_count, d.in_eps = float(in_bn["count"]), float(in_bn.get("eps", 1e-5))
        d.in_stats_replicas = int(in_bn.get("replicas", 0))
    check(lib().st_conv3x3_s2(C.byref(d), _stream()), "st_conv3x3_s2")
    return out


def conv1x1_wreg_supported(Cin, N):
    return int(lib().st_conv1x1_wreg_supported(Cin, N))


def conv1x1_wreg(x, w_frag, N, stride=1, stats=None, stats_replicas=0, scale=None, shift=None, relu=False, residual=None, in_bn=None, out=None,
                 stats_only=False):
    """Register-resident-filter 1x1 conv (st_conv1x1_wreg): x (B,H,W,C) bf16 NHWC, w_frag from pack_conv_weight_frag(w (N,C,1,1)).
    stats_only: y == NULL -- only the [sum | sumsq] statistics of the output are produced (returns None)."""
    _dev(x, w_frag, stats, scale, shift, residual, out)
    B, H, W, Cc = x.shape
    if x.dtype != torch.bfloat16:
        raise _lib.ShowTellHipError("conv1x1_wreg is a bf16 kernel")
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None and not stats_only:
        out = torch.empty(B, Ho, Wo, N, device=x.device, dtype=torch.bfloat16)
    d = Conv1x1WregDesc(_p(x), _p(w_frag), _p(out), _p(residual), _p(stats), int(stats_replicas), _p(scale), _p(shift), int(relu),
                        None, None, None, 0.0, 0.0, 0, B, H, W, Cc, N, int(stride))
    if in_bn is not None:
        _dev(in_bn["stats"], in_bn["gamma"], in_bn["beta"])
        d.in_stats, d.in_gamma, d.in_beta = in_bn["stats"].data_ptr(), in_bn["gamma"].data_ptr(), in_bn["beta"].data_ptr()
        d.in_count, d.in_eps = float(in_bn["count"]), float(in_bn.get("eps", 1e-5))
        d.in_stats_replicas = int(in_bn.get("replicas", 0))
    check(lib().st_conv1x1_wreg(C.byref(d), _stream()), "st_conv1x1_wreg")
    return out


def conv1x1_kfuse_supported(Cin, N):
    return int(lib().st_conv1x1_kfuse_supported(Cin, N))


def conv1x1_kfuse(raw, identity, w_frag, bn, N=256, id_bn=None, stats=None, stats_replicas=0, x_out=None, out=None, eight_waves=False):
    """st_conv1x1_kfuse: x = relu(bn(raw) + identity) (written to x_out; identity normalised with id_bn first when given),
    y = conv1x1(x) (C -> N).  bn / id_bn = dict(stats, gamma, beta[, count, eps, replicas]).  Returns (x_out, y)."""
    _dev(raw, identity, w_frag, stats, x_out, out, bn["stats"], bn["gamma"], bn["beta"])
    rows = raw.numel() // raw.shape[-1]
    if x_out is None:
        x_out = torch.empty_like(raw)
    if out is None:
        out = torch.empty(*raw.shape[:-1], N, device=raw.device, dtype=torch.bfloat16)
    d = Conv1x1KfuseDesc(_p(raw), _p(identity), _p(x_out), _p(w_frag), _p(out), _p(stats), int(stats_replicas), _p(bn["stats"]), _p(bn["gamma"]),
                         _p(bn["beta"]), float(bn["count"]), float(bn.get("eps", 1e-5)), int(bn.get("replicas", 0)), rows, raw.shape[-1], N,
                         None, None, None, 0)
    if id_bn is not None:
        _dev(id_bn["stats"], id_bn["gamma"], id_bn["beta"])
        d.id_stats, d.id_gamma, d.id_beta = id_bn["stats"].data_ptr(), id_bn["gamma"].data_ptr(), id_bn["beta"].data_ptr()
        d.id_stats_replicas = int(id_bn.get("replicas", 0))
    if eight_waves:
        check(lib().st_conv1x1_kfuse8(C.byref(d), _stream()), "st_conv1x1_kfuse8")
    else:
        check(lib().st_conv1x1_kfuse(C.byref(d), _stream()), "st_conv1x1_kfuse")
    return x_out, out


def conv_b2b(raw2, w3_frag, identity, w1_frag, N, bn2, bn3, count, id_bn=None, eps=1e-5, stats=None, stats_replicas=0, x_out=None, out=None):
    """st_conv_b2b: x = relu(bn3(conv3(relu(bn2(raw2)))) + identity) (written to x_out), y = conv1(x) (N == 0: no conv1, y is None).
    bn2 / bn3 / id_bn = dict(stats, gamma, beta[, replicas]).  Returns (x_out, y)."""
    _dev(raw2, w3_frag, identity, w1_frag, stats, x_out, out)
    rows = raw2.numel() // raw2.shape[-1]
    C1, C2 = raw2.shape[-1], identity.shape[-1]
    if x_out is None:
        x_out = torch.empty_like(identity)
    if out is None and N > 0:
        out = torch.empty(*raw2.shape[:-1], N, device=raw2.device, dtype=torch.bfloat16)
    ib = id_bn or {}
    d = ConvB2bDesc(_p(raw2), _p(w3_frag), _p(identity), _p(x_out), _p(w1_frag), _p(out), _p(stats), int(stats_replicas),
                    _p(bn2["stats"]), _p(bn2["gamma"]), _p(bn2["beta"]), int(bn2.get("replicas", 0)),
                    _p(bn3["stats"]), _p(bn3["gamma"]), _p(bn3["beta"]), int(bn3.get("replicas", 0)),
                    _p(ib.get("stats")), _p(ib.get("gamma")), _p(ib.get("beta")), int(ib.get("replicas", 0)),
                    float(count), float(eps), rows, C1, C2, N)
    check(lib().st_conv_b2b(C.byref(d), _stream()), "st_conv_b2b")
    return x_out, out


def conv_c3c1(x2, w3_frag, identity, w1_frag, bn2=None, bn3=None, count=None, eps=1e-5, stats=None, stats_replicas=0,
              scale3=None, shift3=None, scale1=None, shift1=None, relu1=True, x_out=None, out=None, id_bn=None):
    """st_conv_c3c1: x = relu(bn3(conv3(a2)) + identity) (-> x_out), y = conv1_next(x) for the 14 x 14 Bottlenecks (256 -> 1024 -> 256)
    and the 28 x 28 ones (128 -> 512 -> 128).
    train: bn3 = dict(stats, gamma, beta[, replicas]) (+ bn2 for a raw x2), count; eval: scale3 / shift3 / scale1 / shift1.  Returns (x_out, y)."""
    from ._lib import ConvC3c1Desc
    _dev(x2, w3_frag, identity, w1_frag, stats, x_out, out, scale3, shift3, scale1, shift1)
    rows = x2.numel() // x2.shape[-1]
    if x_out is None:
        x_out = torch.empty_like(identity)
    if out is None:
        out = torch.empty(*x2.shape[:-1], x2.shape[-1], device=x2.device, dtype=torch.bfloat16)
    d = ConvC3c1Desc()
    d.x2, d.w3_frag, d.identity, d.x_out, d.w1_frag, d.y = x2.data_ptr(), w3_frag.data_ptr(), identity.data_ptr(), x_out.data_ptr(), w1_frag.data_ptr(), out.data_ptr()
    d.stats, d.stats_replicas = (stats.data_ptr() if stats is not None else None), int(stats_replicas)
    for nm, bn in (("bn2", bn2), ("bn3", bn3), ("id", id_bn)):
        if bn is not None:
            _dev(bn["stats"], bn["gamma"], bn["beta"])
            setattr(d, nm + "_stats", bn["stats"].data_ptr()); setattr(d, nm + "_gamma", bn["gamma"].data_ptr()); setattr(d, nm + "_beta", bn["beta"].data_ptr())
            setattr(d, nm + "_replicas", int(bn.get("replicas", 0)))
    d.count, d.eps = float(count or 0.0), float(eps)
    for nm, t in (("scale3", scale3), ("shift3", shift3), ("scale1", scale1), ("shift1", shift1)):
        if t is not None:
            setattr(d, nm, t.data_ptr())
    d.relu1 = int(relu1)
    d.rows, d.C1, d.C2, d.N = rows, x2.shape[-1], identity.shape[-1], out.shape[-1]
    check(lib().st_conv_c3c1(C.byref(d), _stream()), "st_conv_c3c1")
    return x_out, out


def conv1x1_astat_supported(Cin, N):
    return int(lib().st_conv1x1_astat_supported(Cin, N))


def conv1x1_astat(x, w_frag, N, stride=1, stats=None, stats_replicas=0, scale=None, shift=None, relu=False, in_bn=None, out=None, residual=None,
                  stats_only=False):
    """Activation-stationary 1x1 conv (st_conv1x1_astat): stride 1 with (C, N) in {(256, 1024), (512, 2048)}, stride 2 with (256, 512) /
    (512, 1024); w_frag = pack_conv_weight_frag(w, conv1x1_astat_supported(C, N))."""
    _dev(x, w_frag, stats, scale, shift, out, residual)
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None and not stats_only:                  # stats_only: y == NULL, only the [sum | sumsq] statistics are produced
        out = torch.empty(B, Ho, Wo, N, device=x.device, dtype=torch.bfloat16)
    d = Conv1x1WregDesc(_p(x), _p(w_frag), _p(out), _p(residual), _p(stats), int(stats_replicas), _p(scale), _p(shift), int(relu),
                        None, None, None, 0.0, 0.0, 0, B, H, W, Cc, N, int(stride))
    if in_bn is not None:
        _dev(in_bn["stats"], in_bn["gamma"], in_bn["beta"])
        d.in_stats, d.in_gamma, d.in_beta = in_bn["stats"].data_ptr(), in_bn["gamma"].data_ptr(), in_bn["beta"].data_ptr()
        d.in_count, d.in_eps = float(in_bn["count"]), float(in_bn.get("eps", 1e-5))
        d.in_stats_replicas = int(in_bn.get("replicas", 0))
    check(lib().st_conv1x1_astat(C.byref(d), _stream()), "st_conv1x1_astat")
    return out


def conv1x1_kstream_supported(Cin, N):
    return int(lib().st_conv1x1_kstream_supported(Cin, N))


def conv1x1_kstream(x, w_frag, N, stride=1, stats=None, stats_replicas=0, scale=None, shift=None, relu=False, out=None):
    """Long-K 1x1 conv (st_conv1x1_kstream): x (B,H,W,C) bf16 NHWC with C in {1024, 2048}, w_frag = pack_conv_weight_frag(w, 4)."""
    _dev(x, w_frag, stats, scale, shift, out)
    B, H, W, Cc = x.shape
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    if out is None:
        out = torch.empty(B, Ho, Wo, N, device=x.device, dtype=torch.bfloat16)
    d = Conv1x1WregDesc(_p(x), _p(w_frag), _p(out), None, _p(stats), int(stats_replicas), _p(scale), _p(shift), int(relu),
                        None, None, None, 0.0, 0.0, 0, B, H, W, Cc, N, int(stride))
    check(lib().st_conv1x1_kstream(C.byref(d), _stream()), "st_conv1x1_kstream")
    return out


def pack_conv_weight(w, dtype, cpad=None, k_order=0):
    """(Cout,Cin,KH,KW) fp32 torch layout -> (Cout, KH*KW*Cpad) K-contiguous (k_order: see st_conv_desc)."""
    _dev(w)
    Cout, Cin, KH, KW = w.shape
    cpad = cpad or Cin
    out = torch.empty(Cout, KH * KW * cpad, device=w.device, dtype=dtype)
    check(lib().st_pack_conv_weight(_p(w), _p(out), _DT[dtype], Cout, Cin, KH, KW, cpad, int(k_order), _stream()), "st_pack_conv_weight")
    return out
