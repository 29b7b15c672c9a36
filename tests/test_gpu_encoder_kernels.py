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


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Cc,R", [(64, 4), (96, 1), (256, 3)])
def test_conv_stats_replicas_feed_bn_act(dtype, Cc, R):
    """Replicated batch statistics (st_conv_desc.stats_replicas): the replicas sum to the single-buffer statistics and
    bn_act normalises from them exactly as from one buffer (LDS-table path; C=96 also covers it with R=1)."""
    ops = _ops()
    g = torch.Generator().manual_seed(Cc + R)
    B, H, Cin = 6, 20, 32                                   # M = 2400 -> 19 pixel tiles
    x = torch.randn(B, Cin, H, H, generator=g)
    w = torch.randn(Cc, Cin, 1, 1, generator=g) / np.sqrt(Cin)
    res = torch.randn(B, H, H, Cc, generator=g)
    if dtype == torch.bfloat16:
        x, w, res = x.bfloat16().float(), w.bfloat16().float(), res.bfloat16().float()
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)
    wd = ops.pack_conv_weight(w.cuda(), dtype)
    one = torch.zeros(2 * Cc, device="cuda")
    rep = torch.zeros(R, 2 * Cc, device="cuda")
    y1 = ops.conv_nhwc(xd, wd, 1, 1, 1, 0, stats=one)
    y2 = ops.conv_nhwc(xd, wd, 1, 1, 1, 0, stats=rep, stats_replicas=R)
    assert torch.equal(y1, y2)
    if R > 1:
        assert (rep.abs().sum(1) > 0).all(), "every replica receives tiles"
    np.testing.assert_allclose(rep.sum(0).cpu().numpy(), one.cpu().numpy(), rtol=2e-5, atol=2e-3)
    gam, bet = (torch.rand(Cc, generator=g) + 0.5).cuda(), torch.randn(Cc, generator=g).cuda()
    n = B * H * H
    rd = res.to("cuda", dtype)
    a = ops.bn_act(y1, gam, bet, stats=one, count=n, relu=True, res=rd)
    b = ops.bn_act(y1, gam, bet, stats=rep, count=n, relu=True, res=rd, stats_replicas=R)
    _close(b, a.float().cpu(), dtype, "bn_act from replicas")
    # and against torch
    yr = F.conv2d(x, w)
    ref = F.relu(F.batch_norm(yr, None, None, gam.cpu(), bet.cpu(), True, 0.1, 1e-5).permute(0, 2, 3, 1) + res)
    _close(b, ref, dtype, "conv+bn(train)+res+relu")
    # residual branch with its own replicated statistics
    c = ops.bn_act(y1, gam, bet, stats=one, count=n, relu=False, res=y1, res_bn=dict(gamma=gam, beta=bet, stats=one))
    d = ops.bn_act(y1, gam, bet, stats=rep, count=n, relu=False, res=y1, stats_replicas=R,
                   res_bn=dict(gamma=gam, beta=bet, stats=rep, stats_replicas=R))
    _close(d, c.float().cpu(), dtype, "bn_act + bn(res) from replicas")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 32, 48), (1, 224, 224)])
def test_stem_space_to_depth_matches_conv7x7(dtype, B, H, W):
    """The blocked-image stem (st_nchw_to_s2d16 + st_stem_weight_s2d + sliding-window st_conv) is the same sum as
    torchvision's conv1 (7x7, stride 2, pad 3; cnn.py:46)."""
    ops = _ops()
    g = torch.Generator().manual_seed(H + W)
    x = torch.randn(B, 3, H, W, generator=g)
    w = torch.randn(64, 3, 7, 7, generator=g) / np.sqrt(147)
    cpad = 8 if dtype == torch.bfloat16 else 4
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    ref = F.conv2d(x, w, None, 2, 3).permute(0, 2, 3, 1).contiguous()
    wpad = torch.zeros(64, cpad, 7, 7)
    wpad[:, :3] = w
    wd = ops.pack_conv_weight(wpad.cuda(), dtype)
    stats = torch.zeros(128, device="cuda")
    y, xs, ws = ops.stem_conv_s2d(x.cuda(), wd, cpad, dtype, stats=stats)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    _close(y, ref, dtype, "s2d stem")
    r2 = ref.reshape(-1, 64)
    _close(stats[:64], r2.sum(0), dtype, "sum")
    _close(stats[64:], (r2 * r2).sum(0), dtype, "sumsq")
    # layout facts: zero border / pad channels, and the blocked pixel content
    xs = xs.float().cpu()
    assert xs[:, :2].abs().max() == 0 and xs[:, -1].abs().max() == 0 and xs[:, :, :2].abs().max() == 0 and xs[:, :, -1].abs().max() == 0
    assert xs[..., 12:].abs().max() == 0
    blk = xs[:, 2:-1, 2:-1, :12].reshape(B, H // 2, W // 2, 2, 2, 3)          # (dy, dx, c)
    want = x.reshape(B, 3, H // 2, 2, W // 2, 2).permute(0, 2, 4, 3, 5, 1)
    assert torch.equal(blk, want.to(dtype).float())


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_maxpool_bn_equals_two_pass(dtype):
    """st_maxpool3x3s2_bn (stem tail of torchvision's resnet: bn1, relu, maxpool) == bn_act then maxpool, bit for bit,
    and matches torch."""
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    B, H, W, Cc = 3, 18, 22, 64
    x = torch.randn(B, Cc, H, W, generator=g) * 2 - 0.3
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    gam, bet = torch.randn(Cc, generator=g), torch.randn(Cc, generator=g)     # negative gammas included
    xl = x.permute(0, 2, 3, 1).contiguous()
    x2 = xl.reshape(-1, Cc)
    stats = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).cuda()
    xd = xl.to("cuda", dtype)
    n = B * H * W
    two = ops.maxpool3x3s2(ops.bn_act(xd, gam.cuda(), bet.cuda(), stats=stats, count=n, relu=True))
    one = ops.maxpool3x3s2_bn(xd, gam.cuda(), bet.cuda(), stats=stats, count=n)
    assert torch.equal(one, two)
    ref = F.max_pool2d(F.relu(F.batch_norm(x, None, None, gam, bet, True, 0.1, 1e-5)), 3, 2, 1).permute(0, 2, 3, 1)
    _close(one, ref, dtype, "bn+relu+maxpool")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", [(3, 14, 256, 1024, 1), (2, 28, 128, 512, 1), (5, 7, 512, 2048, 1), (2, 14, 64, 64, 3), (1, 9, 128, 96, 3)])
def test_conv_input_bn_relu_equals_separate_pass(dtype, case):
    """st_conv_desc.in_stats: the consumer conv applies relu(batchnorm(x)) to the producer's raw output in its loader.
    Must equal bn_act followed by the plain conv (bf16: bit for bit -- same coefficients, same rounding point; fp32:
    to an ulp of the normalised input), including the zero padding of a 3x3 consumer (padding taps stay zero, they
    are not relu(shift))."""
    ops = _ops()
    B, H, Cin, Cout, k = case
    g = torch.Generator().manual_seed(B * H + Cin)
    x = (torch.randn(B, H, H, Cin, generator=g) * 1.5 + 0.3).to("cuda", dtype)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k))
    wd = ops.pack_conv_weight(w.cuda(), dtype)
    gam, bet = (torch.rand(Cin, generator=g) + 0.5).cuda(), (torch.randn(Cin, generator=g) * 0.5).cuda()
    x2 = x.float().reshape(-1, Cin)
    stats = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
    n = B * H * H
    y_sep = ops.conv_nhwc(ops.bn_act(x, gam, bet, stats=stats, count=n, relu=True), wd, k, k, 1, k // 2)
    st = torch.zeros(2 * Cout, device="cuda")
    y_fused = ops.conv_nhwc(x, wd, k, k, 1, k // 2, stats=st, in_bn=dict(stats=stats, gamma=gam, beta=bet, count=n))
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        assert torch.equal(y_sep, y_fused)
    else:
        assert (y_sep - y_fused).abs().max().item() <= 2e-6 * y_sep.abs().max().item()
    ref = F.conv2d(F.relu(F.batch_norm(x.float().cpu().permute(0, 3, 1, 2), None, None, gam.cpu(), bet.cpu(), True, 0.1, 1e-5)).to(dtype).float(),
                   w.to(dtype).float(), None, 1, k // 2).permute(0, 2, 3, 1)
    _close(y_fused, ref, dtype, "conv(relu(bn(x)))")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_grouped_gemm_equals_single_launches(dtype):
    """st_conv_batch: problems of identical shape in one launch (blockIdx.y = member) give exactly what st_conv gives
    one by one (the decoder's per-layer weight-gradient GEMMs), including accumulation into the fp32 output."""
    ops = _ops()
    g = torch.Generator().manual_seed(21)
    M, N, K, n = 1536, 512, 1864, 5
    As = [torch.randn(M, K, generator=g).to("cuda", dtype) for _ in range(n)]
    Ws = [(torch.randn(N, K, generator=g) / np.sqrt(K)).to("cuda", dtype) for _ in range(n)]
    base = [torch.randn(M, N, generator=g).cuda() for _ in range(n)]
    one = [ops.gemm_nt(a, w, out_dtype=torch.float32, out=b.clone(), accumulate=True) for a, w, b in zip(As, Ws, base)]
    grp = ops.gemm_nt_batch(As, Ws, [b.clone() for b in base], accumulate=True)
    torch.cuda.synchronize()
    for o, q, a, w, b in zip(one, grp, As, Ws, base):
        assert torch.equal(o, q)
        _close(q, b.cpu() + a.float().cpu() @ w.float().cpu().t(), dtype, "grouped gemm")
    with pytest.raises(Exception):
        ops.gemm_nt_batch(As[:2], [Ws[0], Ws[1][:256].contiguous()], [base[0].clone(), base[1][:, :256].contiguous()])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("M,N,K,S,acc", [(1857, 512, 10240, 8, False), (128, 512, 2048, 8, False), (77, 96, 1024, 4, True)])
def test_split_k_gemm(dtype, M, N, K, S, acc):
    """st_conv_desc.split_k: K slices of a plain GEMM in one grouped launch, fp32 tiles added with atomics (bias once)."""
    ops = _ops()
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, K, generator=g).to("cuda", dtype)
    w = (torch.randn(N, K, generator=g) / np.sqrt(K)).to("cuda", dtype)
    bias = torch.randn(N, generator=g).cuda()
    base = torch.randn(M, N, generator=g).cuda()
    out = base.clone() if acc else torch.full((M, N), 9.0, device="cuda")
    ops.gemm_nt(a, w, out_dtype=torch.float32, bias=bias, out=out, accumulate=acc, split_k=S)
    ref = a.float().cpu() @ w.float().cpu().t() + bias.cpu() + (base.cpu() if acc else 0)
    _close(out, ref, torch.float32 if dtype == torch.float32 else dtype, "split-K gemm")
    one = ops.gemm_nt(a, w, out_dtype=torch.float32, bias=bias, out=base.clone() if acc else None, accumulate=acc)
    assert (one - out).abs().max().item() <= 1e-3 * ref.abs().max().item()


@pytest.mark.parametrize("case", [(2, 14, 14, 256, 256, 3, 1, 1), (2, 28, 28, 128, 128, 3, 2, 1), (3, 14, 14, 1024, 256, 1, 1, 0), (1, 9, 11, 64, 136, 3, 1, 1)])
def test_experimental_small_block_kernel_matches_conv2d(case):
    """The opt-in LDS-DMA "many small blocks" main loop (ST_IGEMM_S3 / st_tune(1, ..)) computes the same convolution
    and statistics as the default kernel (bf16; padding taps and ragged edges go through the zero word)."""
    ops = _ops()
    from showtell_amd._lib import lib
    B, H, W, Cin, Cout, k, s, p = case
    g = torch.Generator().manual_seed(Cin + Cout + k)
    x = torch.randn(B, Cin, H, W, generator=g).bfloat16().float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) / np.sqrt(Cin * k * k)).bfloat16().float()
    ref = F.conv2d(x, w, None, s, p).permute(0, 2, 3, 1).contiguous()
    xd = x.permute(0, 2, 3, 1).contiguous().to("cuda", torch.bfloat16)
    try:
        for ko in (0, 1) if k > 1 else (0,):
            wd = ops.pack_conv_weight(w.cuda(), torch.bfloat16, k_order=ko)
            lib().st_tune(0, -1, -1)
            s0 = torch.zeros(2 * Cout, device="cuda")
            y0 = ops.conv_nhwc(xd, wd, k, k, s, p, stats=s0, k_order=ko)
            for variant in (4,):                         # the three-buffer form (the single-buffer one: `make EXPERIMENTAL=1` builds only)
                lib().st_tune(variant, -1, -1)
                s1 = torch.zeros(2 * Cout, device="cuda")
                y1 = ops.conv_nhwc(xd, wd, k, k, s, p, stats=s1, k_order=ko)
                torch.cuda.synchronize()
                _close(y1, ref, torch.bfloat16, "small-block conv")
                assert (y0.float() - y1.float()).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
                np.testing.assert_allclose(s1.cpu().numpy(), s0.cpu().numpy(), rtol=2e-3, atol=0.5)
    finally:
        lib().st_tune(0, -1, -1)


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
    for rows, cols, ldy in [(1857, 1536, 1864), (300, 10000, 304), (5, 12, 8), (128, 70, 128)]:
        big = torch.randn(rows, cols, generator=g).to("cuda", dtype)
        cs = torch.full((cols,), 2.0, device="cuda")
        bt = ops.transpose(big, ldy=ldy, colsum=cs)
        assert torch.equal(bt[:, :rows].cpu(), big.cpu().t()) and float(bt[:, rows:].float().abs().max() if ldy > rows else 0) == 0.0
        np.testing.assert_allclose(cs.cpu().numpy(), 2.0 + big.float().sum(0).cpu().numpy(), rtol=1e-4, atol=1e-3)
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


@pytest.mark.parametrize("B,H,W", [(2, 64, 64), (3, 32, 48), (1, 224, 224), (2, 34, 70), (1, 8, 8)])
def test_stem_conv_pool_one_kernel_matches_conv_bn_relu_maxpool(B, H, W):
    """st_stem_conv_pool (conv1 7x7/2 + statistics + 3x3/2 pool in one kernel) against torch's conv2d -> BatchNorm (batch statistics)
    -> ReLU -> MaxPool2d(3, 2, 1) (torchvision resnet's first four modules, cnn.py:46), and BIT FOR BIT against the two-kernel form
    (st_conv on the blocked image, st_maxpool3x3s2_bn).  Train mode pools the RAW output with max where gamma >= 0 and min where
    gamma < 0 and leaves bn1 + relu to the consumer: both signs are in gamma here."""
    ops = _ops()
    g = torch.Generator().manual_seed(3 * H + W)
    x = torch.randn(B, 3, H, W, generator=g).bfloat16().float()
    w = (torch.randn(64, 3, 7, 7, generator=g) / np.sqrt(147)).bfloat16().float()
    gamma = torch.rand(64, generator=g) + 0.5
    gamma[::3] *= -1.0                                   # a third of the channels pool with MIN
    beta = torch.randn(64, generator=g) * 0.3
    wpad = torch.zeros(64, 8, 7, 7); wpad[:, :3] = w
    wd = ops.pack_conv_weight(wpad.cuda(), torch.bfloat16)
    xd = x.cuda()
    # two-kernel form
    s0 = torch.zeros(128, device="cuda")
    raw, _, _ = ops.stem_conv_s2d(xd, wd, 8, torch.bfloat16, stats=s0)
    n = float(B * (H // 2) * (W // 2))
    two = ops.maxpool3x3s2_bn(raw, gamma.cuda(), beta.cuda(), stats=s0, count=n)
    # one kernel: pooled raw values + replicated statistics, then bn1 + relu as a consumer would apply them
    R = 4
    s1 = torch.zeros(R, 128, device="cuda")
    pooled = ops.stem_conv_pool(xd, wd, 8, stats=s1, stats_replicas=R, gamma=gamma.cuda())
    torch.cuda.synchronize()
    tot = s1.sum(0)
    np.testing.assert_allclose(tot.cpu().numpy(), s0.cpu().numpy(), rtol=2e-4, atol=2e-3 * np.sqrt(n))
    one = ops.bn_act(pooled, gamma.cuda(), beta.cuda(), stats=s0, count=n, relu=True)     # same statistics as `two`: bit-exact check
    assert one.shape == two.shape
    assert torch.equal(one, two)
    # against torch fp32
    conv = F.conv2d(x, w, None, 2, 3)
    ref = F.max_pool2d(F.relu(F.batch_norm(conv, None, None, gamma, beta, True, 0.0, 1e-5)), 3, 2, 1).permute(0, 2, 3, 1)
    assert (one.float().cpu() - ref).abs().max().item() <= 3e-2 * ref.abs().max().item()
    # eval: folded scale / shift
    sc, sh = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g) * 0.3
    sc[1::4] *= -1.0
    y_e = ops.stem_conv_pool(xd, wd, 8, scale=sc.cuda(), shift=sh.cuda())
    raw_e, _, _ = ops.stem_conv_s2d(xd, wd, 8, torch.bfloat16, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    assert torch.equal(y_e, ops.maxpool3x3s2(raw_e))
    ref_e = F.max_pool2d(F.relu(conv * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)), 3, 2, 1).permute(0, 2, 3, 1)
    assert (y_e.float().cpu() - ref_e).abs().max().item() <= 3e-2 * ref_e.abs().max().item()
