"""GPU parity of the encoder primitives (C ABI via ctypes) against torch CPU fp32 ops.

fp32 kernels: exact-fp32 MFMA, tolerance 1e-4 relative to the output scale.
bf16 kernels: inputs/weights rounded to bf16 on both sides, fp32 accumulation; the
only difference left is the bf16 rounding of the output (rel 2^-8) -> 1.5e-2.
"""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-4, torch.bfloat16: 1.5e-2}


def _ops():
    from showtell_amd import ops
    return ops


def _close(got, ref, dtype, what=""):
    got, ref = got.float().cpu(), ref.float()
    scale = ref.abs().max().item() + 1e-6
    err = (got - ref).abs().max().item() / scale
    assert err < TOL[dtype], f"{what}: rel err {err:.3e} (scale {scale:.3e})"


CONV_CASES = [
    # B, H, W, Cin, Cout, k, s, p
    (2, 14, 14, 256, 256, 3, 1, 1),
    (2, 28, 28, 128, 128, 3, 2, 1),
    (3, 14, 14, 1024, 256, 1, 1, 0),
    (2, 28, 28, 512, 1024, 1, 2, 0),
    (2, 56, 56, 64, 64, 1, 1, 0),
    (1, 7, 7, 512, 2048, 1, 1, 0),
    (2, 32, 32, 8, 64, 7, 2, 3),     # stem geometry, channels padded 3 -> 8
    (1, 9, 11, 16, 24, 3, 1, 1),     # ragged everything
    (3, 28, 28, 128, 128, 3, 1, 1),  # layer2 3x3
    (2, 56, 56, 64, 64, 3, 1, 1),    # layer1 3x3, 64-channel tile
    (5, 7, 7, 512, 512, 3, 1, 1),    # layer4 3x3, ragged batch
    (200, 7, 7, 64, 256, 3, 1, 1),   # many small images
]


@pytest.mark.parametrize("k_order", [0, 1])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_matches_conv2d(case, dtype, k_order):
    ops = _ops()
    B, H, W, Cin, Cout, k, s, p = case
    if k_order and (k == 1 or Cin % 64):
        pytest.skip("channel-chunk-major K order applies to k>1 convs with Cin % 64 == 0")
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, None, s, p).permute(0, 2, 3, 1).contiguous()
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
    wd = ops.pack_conv_weight(w.cuda(), dtype, k_order=k_order)
    stats = torch.zeros(2 * Cout, device="cuda")
    y = ops.conv_nhwc(xd, wd, k, k, s, p, stats=stats, k_order=k_order)
    torch.cuda.synchronize()
    _close(y, ref, dtype, "conv")
    # batch-norm statistics from the epilogue (fp32 accumulators, before output rounding)
    r2 = ref.reshape(-1, Cout)
    _close(stats[:Cout], r2.sum(0), torch.float32 if dtype == torch.float32 else dtype, "sum")
    _close(stats[Cout:], (r2 * r2).sum(0), torch.float32 if dtype == torch.float32 else dtype, "sumsq")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K", [(1600, 1536, 512), (77, 1000, 64), (128, 512, 2048), (5, 12, 8), (300, 10000, 512)])
def test_gemm_nt_bias_relu_accumulate(M, N, K, dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(M + N + K)
    a = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / np.sqrt(K)
    b = torch.randn(N, generator=g)
    if dtype == torch.bfloat16:
        a, w = a.bfloat16().float(), w.bfloat16().float()
    ref = a @ w.t() + b
    y = ops.gemm_nt(a.to("cuda", dtype), w.to("cuda", dtype), out_dtype=torch.float32, bias=b.cuda())
    _close(y, ref, dtype, "gemm+bias")
    y2 = ops.gemm_nt(a.to("cuda", dtype), w.to("cuda", dtype), out_dtype=torch.float32, bias=b.cuda(), relu=True)
    _close(y2, ref.clamp_min(0), dtype, "gemm+bias+relu")
    acc = torch.ones(M, N, device="cuda")
    ops.gemm_nt(a.to("cuda", dtype), w.to("cuda", dtype), out=acc, accumulate=True)
    _close(acc, a @ w.t() + 1.0, dtype, "gemm accumulate")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_epilogue_affine_residual_relu(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, H, Cin, Cout = 2, 14, 64, 256
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cout, Cin, 1, 1, generator=g) / 8
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g)
    res = torch.randn(B, H, H, Cout, generator=g)
    if dtype == torch.bfloat16:
        x, w, res = x.bfloat16().float(), w.bfloat16().float(), res.bfloat16().float()
    ref = F.relu(F.conv2d(x, w).permute(0, 2, 3, 1) * sc + sh + res)
    y = ops.conv_nhwc(x.permute(0, 2, 3, 1).contiguous().to("cuda", dtype), ops.pack_conv_weight(w.cuda(), dtype),
                      1, 1, 1, 0, scale=sc.cuda(), shift=sh.cuda(), residual=res.to("cuda", dtype), relu=True)
    _close(y, ref, dtype, "folded-BN epilogue")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("train", [True, False])
def test_bn_act_residual_variants(dtype, train):
    ops = _ops()
    g = torch.Generator().manual_seed(5)
    B, H, Cc = 3, 7, 64
    x = torch.randn(B, Cc, H, H, generator=g) * 2 + 0.5
    r = torch.randn(B, Cc, H, H, generator=g)
    if dtype == torch.bfloat16:
        x, r = x.bfloat16().float(), r.bfloat16().float()
    gam, bet = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    gam2, bet2 = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    rm, rv = torch.randn(Cc, generator=g) * 0.1, torch.rand(Cc, generator=g) + 0.5
    xl = x.permute(0, 2, 3, 1).contiguous()
    rl = r.permute(0, 2, 3, 1).contiguous()
    n = B * H * H

    def stats_of(t):
        t2 = t.reshape(-1, Cc)
        return torch.cat([t2.sum(0), (t2 * t2).sum(0)]).cuda()

    def bn(t, gg, bb):
        return F.batch_norm(t, rm.clone(), rv.clone(), gg, bb, train, 0.1, 1e-5)

    kw = dict(stats=stats_of(xl), count=n) if train else dict(running=(rm.cuda(), rv.cuda()))
    xd, rd = xl.to("cuda", dtype), rl.to("cuda", dtype)
    # plain bn + relu
    _close(ops.bn_act(xd, gam.cuda(), bet.cuda(), relu=True, **kw).permute(0, 3, 1, 2), F.relu(bn(x, gam, bet)), dtype, "bn+relu")
    # bn + identity residual + relu
    _close(ops.bn_act(xd, gam.cuda(), bet.cuda(), relu=True, res=rd, **kw).permute(0, 3, 1, 2),
           F.relu(bn(x, gam, bet) + r), dtype, "bn+res+relu")
    # bn + bn(residual) + relu (downsample branch)
    rkw = dict(stats=stats_of(rl)) if train else dict(running=(rm.cuda(), rv.cuda()))
    y = ops.bn_act(xd, gam.cuda(), bet.cuda(), relu=True, res=rd,
                   res_bn=dict(gamma=gam2.cuda(), beta=bet2.cuda(), **rkw), **kw)
    _close(y.permute(0, 3, 1, 2), F.relu(bn(x, gam, bet) + bn(r, gam2, bet2)), dtype, "bn+bn(res)+relu")


def test_bn_update_running_matches_torch():
    ops = _ops()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(6, 32, 5, 5, generator=g) * 3 + 1
    rm, rv = torch.zeros(32), torch.ones(32)
    F.batch_norm(x, rm, rv, None, None, True, 0.1, 1e-5)
    x2 = x.permute(0, 2, 3, 1).reshape(-1, 32)
    stats = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).cuda()
    drm, drv = torch.zeros(32, device="cuda"), torch.ones(32, device="cuda")
    ops.bn_update_running(stats, drm, drv, x2.shape[0], 0.1)
    np.testing.assert_allclose(drm.cpu().numpy(), rm.numpy(), atol=1e-5)
    np.testing.assert_allclose(drv.cpu().numpy(), rv.numpy(), rtol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_layout_and_pool_kernels(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(3, 3, 20, 18, generator=g)
    y = ops.nchw_to_nhwc(x.cuda(), dtype, 8)
    assert y.shape == (3, 20, 18, 8)
    _close(y[..., :3], x.permute(0, 2, 3, 1), dtype, "nchw->nhwc")
    assert float(y[..., 3:].abs().max()) == 0.0
    a = torch.randn(2, 64, 15, 13, generator=g)
    al = a.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
    af = al.float().cpu().permute(0, 3, 1, 2)
    _close(ops.maxpool3x3s2(al).permute(0, 3, 1, 2), F.max_pool2d(af, 3, 2, 1), dtype, "maxpool")
    _close(ops.global_avgpool(al, torch.float32), af.mean((2, 3)), dtype, "avgpool")
    f = torch.randn(2, 128, 7, 7, generator=g)
    fl = f.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
    got = ops.nhwc_to_ncp_f32(fl)
    assert got.shape == (2, 128, 49)
    _close(got, fl.float().cpu().permute(0, 3, 1, 2).reshape(2, 128, 49), torch.float32, "nhwc->ncp")
    t = torch.randn(77, 130, generator=g).to("cuda", dtype)
    tt = ops.transpose(t, ldy=80)
    assert tt.shape == (130, 80)
    assert torch.equal(tt[:, :77].cpu(), t.cpu().t())
    assert float(tt[:, 77:].float().abs().max()) == 0.0
    c = ops.cast(t, torch.float32)
    assert torch.equal(c.cpu(), t.float().cpu())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv3x3_eval_epilogue_chunk_major_weights(dtype):
    """3x3 conv, chunk-major weights, folded-BN + residual + ReLU epilogue (BasicBlock conv2 in eval mode)."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    B, H, Cc = 3, 14, 64
    x = torch.randn(B, Cc, H, H, generator=g)
    w = torch.randn(Cc, Cc, 3, 3, generator=g) / 24
    sc, sh = torch.rand(Cc, generator=g) + 0.5, torch.randn(Cc, generator=g)
    res = torch.randn(B, H, H, Cc, generator=g)
    if dtype == torch.bfloat16:
        x, w, res = x.bfloat16().float(), w.bfloat16().float(), res.bfloat16().float()
    ref = F.relu(F.conv2d(x, w, None, 1, 1).permute(0, 2, 3, 1) * sc + sh + res)
    y = ops.conv_nhwc(x.permute(0, 2, 3, 1).contiguous().to("cuda", dtype), ops.pack_conv_weight(w.cuda(), dtype, k_order=1),
                      3, 3, 1, 1, scale=sc.cuda(), shift=sh.cuda(), residual=res.to("cuda", dtype), relu=True, k_order=1)
    _close(y, ref, dtype, "3x3 folded-BN epilogue")


def test_bad_arguments_fail_loudly():
    from showtell_amd import ShowTellHipError
    ops = _ops()
    x = torch.zeros(1, 4, 4, 6, device="cuda", dtype=torch.bfloat16)   # Cin not a multiple of 8
    w = torch.zeros(8, 6, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(ShowTellHipError):
        ops.conv_nhwc(x, w, 1, 1, 1, 0)
    with pytest.raises(ShowTellHipError):
        ops.cast(torch.zeros(4), torch.bfloat16)                        # CPU tensor: no fallback
