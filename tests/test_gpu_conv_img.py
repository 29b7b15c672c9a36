"""st_conv3x3_img (csrc/conv_img.hip): the image-resident 3x3 convolution against torch's conv2d on the same bf16-rounded
operands and against the implicit-GEMM kernel (st_conv) it replaces on those layers -- every ResNet bottleneck width,
bands that split an image (28x28, 56x56), a ragged last band, tiny maps, with and without the fused input
BatchNorm + ReLU, train-mode statistics and the eval-mode scale/shift/ReLU epilogue."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (B, H, W, C, N)
CASES = [
    (3, 14, 14, 256, 256),     # layer3 conv2: two 7-row bands per image, two workgroups per CU
    (2, 13, 11, 256, 256),     # the same form, ragged: bands of 8 + 5 rows at Wp = 13
    (2, 20, 14, 256, 128),     # ... three bands (7 + 7 + 6 rows), one channel half
    (2, 28, 28, 128, 128),     # layer2 conv2: 4 bands of 7 rows
    (2, 56, 56, 64, 64),       # layer1 conv2: 14 bands of 4 rows, all filters in registers
    (5, 7, 7, 512, 512),       # layer4 conv2
    (2, 30, 30, 128, 256),     # ragged last band (7 + 7 + 7 + 7 + 2 rows), N > C
    (3, 4, 4, 256, 128),       # tiny map (a 64x64 input image)
    (1, 9, 5, 64, 128),        # non-square
    (2, 16, 16, 512, 128),     # several bands at 512 channels
]


def _ops():
    from showtell_amd import ops
    return ops


def _data(case, seed=0):
    B, H, W, C, N = case
    g = torch.Generator().manual_seed(1000 * seed + H * W + C)
    x = (torch.randn(B, H, W, C, generator=g) * 1.2 + 0.2).bfloat16()
    w = (torch.randn(N, C, 3, 3, generator=g) / np.sqrt(9 * C)).bfloat16().float()
    return x, w


@pytest.mark.parametrize("case", CASES)
def test_conv3x3_img_matches_conv2d_and_igemm(case):
    ops = _ops()
    B, H, W, C, N = case
    ntw = ops.conv3x3_img_supported(H, W, C, N)
    assert ntw > 0
    x, w = _data(case)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, 1, 1).permute(0, 2, 3, 1).contiguous()
    xd = x.cuda()
    wf = ops.pack_conv_weight_frag(w.cuda(), ntw)
    st = torch.zeros(2 * N, device="cuda")
    y = ops.conv3x3_img(xd, wf, N, stats=st)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * scale
    # the implicit-GEMM kernel on the same operands: both round ONE fp32 sum to bf16; sums differ only by association
    s0 = torch.zeros(2 * N, device="cuda")
    y0 = ops.conv_nhwc(xd, ops.pack_conv_weight(w.cuda(), torch.bfloat16), 3, 3, 1, 1, stats=s0)
    d = (y.float() - y0.float()).abs().max().item()
    assert d <= 2.0 ** -7 * scale, d
    assert (y != y0).float().mean().item() < 0.02       # a different summation order flips the last bit of a few outputs
    r2 = ref.reshape(-1, N)
    np.testing.assert_allclose(st[:N].cpu().numpy(), r2.sum(0).numpy(), rtol=2e-3, atol=2e-3 * scale * np.sqrt(r2.shape[0]))
    np.testing.assert_allclose(st[N:].cpu().numpy(), (r2 * r2).sum(0).numpy(), rtol=2e-3)
    np.testing.assert_allclose(st.cpu().numpy(), s0.cpu().numpy(), rtol=1e-4, atol=1e-3 * scale * np.sqrt(r2.shape[0]))


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[2], CASES[4], CASES[7]])
def test_conv3x3_img_fused_input_bn_relu_equals_separate_pass(case):
    """in_stats: the fill applies relu(batchnorm(x)) once per element; must equal bn_act followed by the plain kernel BIT
    FOR BIT (same coefficients, same rounding point), halo included (padding stays zero, it is not relu(shift))."""
    ops = _ops()
    B, H, W, C, N = case
    ntw = ops.conv3x3_img_supported(H, W, C, N)
    x, w = _data(case, seed=1)
    xd = x.cuda()
    wf = ops.pack_conv_weight_frag(w.cuda(), ntw)
    g = torch.Generator().manual_seed(7)
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    x2 = xd.float().reshape(-1, C)
    stats = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
    n = float(B * H * W)
    y_sep = ops.conv3x3_img(ops.bn_act(xd, gam, bet, stats=stats, count=n, relu=True), wf, N)
    s1 = torch.zeros(4, 2 * N, device="cuda")
    y_fused = ops.conv3x3_img(xd, wf, N, stats=s1, stats_replicas=4, in_bn=dict(stats=stats, gamma=gam, beta=bet, count=n))
    # replicated producer statistics ([R][2C], summed by the kernel's prologue): [stats, 0, 0] adds up exactly
    rep = torch.zeros(3, 2 * C, device="cuda"); rep[1] = stats
    y_rep = ops.conv3x3_img(xd, wf, N, in_bn=dict(stats=rep, gamma=gam, beta=bet, count=n, replicas=3))
    torch.cuda.synchronize()
    assert torch.equal(y_sep, y_fused)
    assert torch.equal(y_rep, y_fused)
    xn = F.relu(F.batch_norm(x.float().permute(0, 3, 1, 2), None, None, gam.cpu(), bet.cpu(), True, 0.1, 1e-5)).bfloat16().float()
    ref = F.conv2d(xn, w, None, 1, 1).permute(0, 2, 3, 1)
    assert (y_fused.float().cpu() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    tot = s1.sum(0).cpu()                                  # replicated statistics add up to the single-buffer ones
    r2 = ref.reshape(-1, N)
    np.testing.assert_allclose(tot[:N].numpy(), r2.sum(0).numpy(), rtol=5e-3, atol=5e-3 * ref.abs().max().item() * np.sqrt(r2.shape[0]))


def test_conv3x3_img_eval_epilogue_and_errors():
    ops = _ops()
    from showtell_amd import ShowTellHipError
    case = (2, 14, 14, 256, 256)
    B, H, W, C, N = case
    x, w = _data(case, seed=2)
    wf = ops.pack_conv_weight_frag(w.cuda(), ops.conv3x3_img_supported(H, W, C, N))
    g = torch.Generator().manual_seed(3)
    sc, sh = (torch.rand(N, generator=g) + 0.5), torch.randn(N, generator=g) * 0.3
    y = ops.conv3x3_img(x.cuda(), wf, N, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w, None, 1, 1).permute(0, 2, 3, 1) * sc + sh)
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    assert ops.conv3x3_img_supported(14, 14, 96, 128) == 0          # no kernel for that width: callers fall back to st_conv
    assert ops.conv3x3_img_supported(300, 300, 512, 512) == 0       # a band of one row would not fit in LDS
    with pytest.raises(ShowTellHipError):
        ops.conv3x3_img(torch.zeros(1, 14, 14, 96, device="cuda", dtype=torch.bfloat16), wf, 128)


# ---- st_conv1x1_wreg: (B, H, W, C, N, stride) -----------------------------------------------------------------------
PW_CASES = [
    (3, 14, 14, 256, 1024, 1),    # layer3 conv3
    (2, 56, 56, 64, 256, 1),      # layer1 conv3 / downsample
    (2, 56, 56, 64, 64, 1),       # layer1 conv1 of block 0
    (2, 56, 56, 256, 64, 1),      # layer1 conv1
    (2, 28, 28, 128, 512, 1),     # layer2 conv3
    (2, 28, 28, 512, 128, 1),     # layer2 conv1
    (3, 28, 28, 512, 1024, 2),    # layer3 downsample (stride 2)
    (2, 56, 56, 256, 512, 2),     # layer2 downsample
    (5, 7, 7, 512, 2048, 1),      # layer4 conv3, ragged rows
    (1, 5, 3, 256, 128, 1),       # a single ragged stage
    (2, 9, 9, 256, 128, 2),       # odd map, stride 2
]


def _pw_data(case, seed=0):
    B, H, W, C, N, s = case
    g = torch.Generator().manual_seed(77 * seed + H * W + C + N)
    x = (torch.randn(B, H, W, C, generator=g) * 1.2 + 0.2).bfloat16()
    w = (torch.randn(N, C, 1, 1, generator=g) / np.sqrt(C)).bfloat16().float()
    return x, w


@pytest.mark.parametrize("case", PW_CASES)
def test_conv1x1_wreg_matches_conv2d_and_igemm(case):
    ops = _ops()
    B, H, W, C, N, s = case
    ntw = ops.conv1x1_wreg_supported(C, N)
    assert ntw > 0
    x, w = _pw_data(case)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, 0).permute(0, 2, 3, 1).contiguous()
    xd = x.cuda()
    wf = ops.pack_conv_weight_frag(w.cuda(), ntw)
    R = 4
    st = torch.zeros(R, 2 * N, device="cuda")
    y = ops.conv1x1_wreg(xd, wf, N, stride=s, stats=st, stats_replicas=R)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert y.shape == ref.shape
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * scale
    s0 = torch.zeros(2 * N, device="cuda")
    y0 = ops.conv_nhwc(xd, ops.pack_conv_weight(w.cuda(), torch.bfloat16), 1, 1, s, 0, stats=s0)
    assert (y.float() - y0.float()).abs().max().item() <= 2.0 ** -7 * scale
    assert (y != y0).float().mean().item() < 0.02
    r2 = ref.reshape(-1, N)
    tot = st.sum(0).cpu().numpy()
    np.testing.assert_allclose(tot[:N], r2.sum(0).numpy(), rtol=2e-3, atol=2e-3 * scale * np.sqrt(r2.shape[0]))
    np.testing.assert_allclose(tot[N:], (r2 * r2).sum(0).numpy(), rtol=2e-3)
    np.testing.assert_allclose(tot, s0.cpu().numpy(), rtol=1e-4, atol=1e-3 * scale * np.sqrt(r2.shape[0]))


@pytest.mark.parametrize("case", [PW_CASES[0], PW_CASES[1], PW_CASES[4], PW_CASES[5], PW_CASES[8], PW_CASES[9]])
def test_conv1x1_wreg_fused_input_bn_relu_and_eval_epilogue(case):
    ops = _ops()
    B, H, W, C, N, s = case
    ntw = ops.conv1x1_wreg_supported(C, N)
    x, w = _pw_data(case, seed=1)
    xd = x.cuda()
    wf = ops.pack_conv_weight_frag(w.cuda(), ntw)
    g = torch.Generator().manual_seed(9)
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    x2 = xd.float().reshape(-1, C)
    stats = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
    n = float(B * H * W)
    y_sep = ops.conv1x1_wreg(ops.bn_act(xd, gam, bet, stats=stats, count=n, relu=True), wf, N, stride=s)
    rep = torch.zeros(3, 2 * C, device="cuda"); rep[2] = stats
    y_fused = ops.conv1x1_wreg(xd, wf, N, stride=s, in_bn=dict(stats=rep, gamma=gam, beta=bet, count=n, replicas=3))
    torch.cuda.synchronize()
    assert torch.equal(y_sep, y_fused)                      # same coefficients, same rounding point as the separate pass
    # eval-mode epilogue: scale / shift, ReLU (the residual form of the Bottleneck's last conv stays with st_conv)
    sc, sh = (torch.rand(N, generator=g) + 0.5), torch.randn(N, generator=g) * 0.3
    y = ops.conv1x1_wreg(xd, wf, N, stride=s, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    ref = F.relu(F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, 0).permute(0, 2, 3, 1) * sc + sh)
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()
    from showtell_amd import ShowTellHipError
    with pytest.raises(ShowTellHipError):
        ops.conv1x1_wreg(xd, wf, N, stride=s, residual=y)        # the residual form needs the eval-mode scale / shift
    # eval-mode conv3 of a Bottleneck (stride 1): relu(conv * scale + shift + identity), the identity requested D stages ahead
    conv = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, 0).permute(0, 2, 3, 1)
    idn = (torch.randn(conv.shape, generator=g) * conv.abs().mean().item() * 2).bfloat16()
    for relu in (True, False):
        y3 = ops.conv1x1_wreg(xd, wf, N, stride=s, scale=sc.cuda(), shift=sh.cuda(), relu=relu, residual=idn.cuda())
        ref3 = conv * sc + sh + idn.float()
        ref3 = F.relu(ref3) if relu else ref3
        assert (y3.float().cpu() - ref3).abs().max().item() <= 1.5e-2 * ref3.abs().max().item()
    y0r = ops.conv_nhwc(xd, ops.pack_conv_weight(w.cuda(), torch.bfloat16), 1, 1, s, 0, scale=sc.cuda(), shift=sh.cuda(), relu=False, residual=idn.cuda())
    assert (y3.float() - y0r.float()).abs().max().item() <= 2.0 ** -7 * ref3.abs().max().item()


# ---- st_conv1x1_kstream: (B, H, W, C, N, stride) ---------------------------------------------------------------------
KS_CASES = [
    (4, 14, 14, 1024, 256, 1),    # layer3 conv1
    (3, 14, 14, 1024, 512, 1),    # layer4 conv1 of block 0 (two channel slices)
    (5, 7, 7, 2048, 512, 1),      # layer4 conv1, ragged rows (245 = 2 x 112 + 21)
    (3, 14, 14, 1024, 2048, 2),   # layer4 downsample (stride 2)
    (1, 3, 5, 1024, 256, 1),      # fewer rows than one tile
]


@pytest.mark.parametrize("case", KS_CASES)
def test_conv1x1_kstream_matches_conv2d_and_igemm(case):
    ops = _ops()
    B, H, W, C, N, s = case
    ntw = ops.conv1x1_kstream_supported(C, N)          # 4: the 1024 -> 256 conv1s (the packing st_conv_c3c1 indexes); 2: the two-per-CU form of the others
    assert ntw == (4 if N == 256 else ntw) and ntw in (2, 4) and ops.conv1x1_kstream_supported(512, 256) == 0
    x, w = _pw_data(case)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, s, 0).permute(0, 2, 3, 1).contiguous()
    xd = x.cuda()
    wf = ops.pack_conv_weight_frag(w.cuda(), ntw)
    R = 4
    st = torch.zeros(R, 2 * N, device="cuda")
    y = ops.conv1x1_kstream(xd, wf, N, stride=s, stats=st, stats_replicas=R)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert y.shape == ref.shape
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * scale
    s0 = torch.zeros(2 * N, device="cuda")
    y0 = ops.conv_nhwc(xd, ops.pack_conv_weight(w.cuda(), torch.bfloat16), 1, 1, s, 0, stats=s0)
    assert (y.float() - y0.float()).abs().max().item() <= 2.0 ** -7 * scale
    assert (y != y0).float().mean().item() < 0.02
    r2 = ref.reshape(-1, N)
    tot = st.sum(0).cpu().numpy()
    np.testing.assert_allclose(tot[:N], r2.sum(0).numpy(), rtol=2e-3, atol=2e-3 * scale * np.sqrt(r2.shape[0]))
    np.testing.assert_allclose(tot[N:], (r2 * r2).sum(0).numpy(), rtol=2e-3)
    # eval-mode epilogue
    g = torch.Generator().manual_seed(4)
    sc, sh = (torch.rand(N, generator=g) + 0.5), torch.randn(N, generator=g) * 0.3
    y2 = ops.conv1x1_kstream(xd, wf, N, stride=s, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    ref2 = F.relu(ref * sc + sh)
    assert (y2.float().cpu() - ref2).abs().max().item() <= 1.5e-2 * ref2.abs().max().item()


# ---- st_conv1x1_astat: (B, H, W, C, N) -----------------------------------------------------------------------------------
AS_CASES = [(4, 14, 14, 256, 1024, 1), (5, 7, 7, 512, 2048, 1), (1, 3, 5, 256, 1024, 1),
            (74, 14, 14, 512, 2048, 1),   # > 128 row blocks: one workgroup per row block walks all 2048 channels (fewer: four channel parts)
            (3, 28, 28, 256, 512, 2), (2, 28, 28, 512, 1024, 2), (5, 9, 7, 256, 512, 2), (1, 1, 3, 512, 1024, 2)]   # stride 2: the downsample convs


@pytest.mark.parametrize("case", AS_CASES)
def test_conv1x1_astat_matches_conv2d_igemm_and_separate_bn_pass(case):
    ops = _ops()
    B, H, W, C, N, S = case
    assert ops.conv1x1_astat_supported(C, N) == 2 and ops.conv1x1_astat_supported(256, 256) == 0
    x, w = _pw_data(case)
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, S, 0).permute(0, 2, 3, 1).contiguous()
    xd = x.cuda()
    wf = ops.pack_conv_weight_frag(w.cuda(), ops.conv1x1_astat_supported(C, N))
    R = 4
    st = torch.zeros(R, 2 * N, device="cuda")
    y = ops.conv1x1_astat(xd, wf, N, stride=S, stats=st, stats_replicas=R)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * scale
    s0 = torch.zeros(2 * N, device="cuda")
    y0 = ops.conv_nhwc(xd, ops.pack_conv_weight(w.cuda(), torch.bfloat16), 1, 1, S, 0, stats=s0)
    assert (y.float() - y0.float()).abs().max().item() <= 2.0 ** -7 * scale
    assert (y != y0).float().mean().item() < 0.02
    r2 = ref.reshape(-1, N)
    tot = st.sum(0).cpu().numpy()
    np.testing.assert_allclose(tot[:N], r2.sum(0).numpy(), rtol=2e-3, atol=2e-3 * scale * np.sqrt(r2.shape[0]))
    np.testing.assert_allclose(tot[N:], (r2 * r2).sum(0).numpy(), rtol=2e-3)
    # fused producer BatchNorm + ReLU == separate pass, bit for bit; replicated producer statistics
    g = torch.Generator().manual_seed(11)
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    x2 = xd.float().reshape(-1, C)
    stats = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
    n = float(B * H * W)
    y_sep = ops.conv1x1_astat(ops.bn_act(xd, gam, bet, stats=stats, count=n, relu=True), wf, N, stride=S)
    rep = torch.zeros(3, 2 * C, device="cuda"); rep[0] = stats
    y_fused = ops.conv1x1_astat(xd, wf, N, stride=S, in_bn=dict(stats=rep, gamma=gam, beta=bet, count=n, replicas=3))
    assert torch.equal(y_sep, y_fused)
    # eval-mode epilogue
    sc, sh = (torch.rand(N, generator=g) + 0.5), torch.randn(N, generator=g) * 0.3
    y2 = ops.conv1x1_astat(xd, wf, N, stride=S, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    ref2 = F.relu(ref * sc + sh)
    assert (y2.float().cpu() - ref2).abs().max().item() <= 1.5e-2 * ref2.abs().max().item()
    if S == 1:
        # eval-mode conv3 of a Bottleneck: relu(conv * scale + shift + identity) in the epilogue (the identity requested a chunk ahead),
        # against fp32 and against the implicit-GEMM kernel's residual epilogue (same arithmetic order: fma, add, max)
        idn = (torch.randn(ref.shape, generator=g) * ref.abs().mean().item() * 2).bfloat16()
        y3 = ops.conv1x1_astat(xd, wf, N, scale=sc.cuda(), shift=sh.cuda(), relu=True, residual=idn.cuda())
        ref3 = F.relu(ref * sc + sh + idn.float())
        assert (y3.float().cpu() - ref3).abs().max().item() <= 1.5e-2 * ref3.abs().max().item()
        y3n = ops.conv1x1_astat(xd, wf, N, scale=sc.cuda(), shift=sh.cuda(), relu=False, residual=idn.cuda())
        ref3n = ref * sc + sh + idn.float()
        assert (y3n.float().cpu() - ref3n).abs().max().item() <= 1.5e-2 * ref3n.abs().max().item()
        y0r = ops.conv_nhwc(xd, ops.pack_conv_weight(w.cuda(), torch.bfloat16), 1, 1, 1, 0, scale=sc.cuda(), shift=sh.cuda(), relu=True, residual=idn.cuda())
        assert (y3.float() - y0r.float()).abs().max().item() <= 2.0 ** -7 * ref3.abs().max().item()
        with pytest.raises(Exception):
            ops.conv1x1_astat(xd, wf, N, relu=False, residual=idn.cuda())          # the residual form needs scale / shift


@pytest.mark.parametrize("eight", [False, True])
@pytest.mark.parametrize("rows_shape", [(4, 14, 14), (1, 9, 13), (3, 7, 5)])
def test_conv1x1_kfuse_equals_bn_act_then_kstream(rows_shape, eight):
    """st_conv1x1_kfuse = the block-end normalise pass (relu(bn3(raw) + identity)) + conv1 of the next block in one kernel: x_out must
    equal st_bn_act's output bit for bit (same coefficients, same rounding point), y the K-streaming kernel's on that x."""
    ops = _ops()
    B, H, W = rows_shape
    C, N = 1024, 256
    from showtell_amd import _lib
    if ops.conv1x1_kfuse_supported(C, N) == 0 or (eight and not _lib.has_symbol("st_conv1x1_kfuse8")):
        pytest.skip("the 1024-channel fused forms are measured-and-unrouted kernels: `make EXPERIMENTAL=1` builds only")
    g = torch.Generator().manual_seed(B * H + W)
    raw = (torch.randn(B, H, W, C, generator=g) * 1.3 + 0.1).bfloat16().cuda()
    ident = torch.relu(torch.randn(B, H, W, C, generator=g)).bfloat16().cuda()
    w = (torch.randn(N, C, 1, 1, generator=g) / np.sqrt(C)).bfloat16().float()
    wf = ops.pack_conv_weight_frag(w.cuda(), 4)
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    r2 = raw.float().reshape(-1, C)
    stats = torch.cat([r2.sum(0), (r2 * r2).sum(0)]).contiguous()
    rep = torch.zeros(4, 2 * C, device="cuda"); rep[3] = stats
    n = float(B * H * W)
    x_ref = ops.bn_act(raw, gam, bet, stats=stats, count=n, relu=True, res=ident)
    s0 = torch.zeros(2 * N, device="cuda")
    y_ref = ops.conv1x1_kstream(x_ref, wf, N, stats=s0)
    s1 = torch.zeros(2, 2 * N, device="cuda")
    x, y = ops.conv1x1_kfuse(raw, ident, wf, dict(stats=rep, gamma=gam, beta=bet, count=n, replicas=4), stats=s1, stats_replicas=2, eight_waves=eight)
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref)
    assert torch.equal(y, y_ref)
    np.testing.assert_allclose(s1.sum(0).cpu().numpy(), s0.cpu().numpy(), rtol=1e-4, atol=1e-2)
    ref = F.conv2d(x_ref.float().cpu().permute(0, 3, 1, 2), w, None, 1, 0).permute(0, 2, 3, 1)
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()


# (rows shape, C, N, identity needs its own BatchNorm): the fused loader of the register-resident-filter kernel (layer1 / layer2 widths)
KF_PW_CASES = [((2, 56, 56), 256, 64, False), ((3, 28, 28), 256, 128, False), ((2, 28, 28), 512, 128, False), ((2, 28, 28), 512, 256, False),
               ((1, 56, 56), 256, 64, True), ((3, 9, 7), 512, 128, True), ((1, 1, 5), 256, 128, False)]


@pytest.mark.parametrize("case", KF_PW_CASES)
def test_conv1x1_kfuse_register_filter_form_equals_bn_act_then_wreg(case):
    """C = 256 / 512: x_out == st_bn_act(raw, res = identity[, res_bn]) bit for bit, y == st_conv1x1_wreg on that x bit for bit, for
    every channel-slice count (N = 64 .. 256: 1 - 4 slices read the same rows, slice 0 writes x_out)."""
    ops = _ops()
    shape, C, N, idbn = case
    B, H, W = shape
    ntw = ops.conv1x1_kfuse_supported(C, N)
    assert ntw > 0 and ntw == ops.conv1x1_wreg_supported(C, N)
    g = torch.Generator().manual_seed(B * H + W + C + N)
    raw = (torch.randn(B, H, W, C, generator=g) * 1.3 + 0.1).bfloat16().cuda()
    ident = (torch.randn(B, H, W, C, generator=g) * (1.0 if idbn else 0.7)).bfloat16().cuda()
    if not idbn:
        ident = torch.relu(ident)
    w = (torch.randn(N, C, 1, 1, generator=g) / np.sqrt(C)).bfloat16().float()
    wf = ops.pack_conv_weight_frag(w.cuda(), ntw)
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    gam2, bet2 = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    n = float(B * H * W)

    def stats_of(t):
        t2 = t.float().reshape(-1, C)
        return torch.cat([t2.sum(0), (t2 * t2).sum(0)]).contiguous()
    st_raw, st_id = stats_of(raw), stats_of(ident)
    rep = torch.zeros(4, 2 * C, device="cuda"); rep[2] = st_raw
    rep_id = torch.zeros(2, 2 * C, device="cuda"); rep_id[1] = st_id
    kw = dict(res=ident)
    if idbn:
        kw.update(res_bn=dict(stats=st_id, gamma=gam2, beta=bet2))
    x_ref = ops.bn_act(raw, gam, bet, stats=st_raw, count=n, relu=True, **kw)
    s0 = torch.zeros(2 * N, device="cuda")
    y_ref = ops.conv1x1_wreg(x_ref, wf, N, stats=s0)
    s1 = torch.zeros(2, 2 * N, device="cuda")
    x, y = ops.conv1x1_kfuse(raw, ident, wf, dict(stats=rep, gamma=gam, beta=bet, count=n, replicas=4), N=N,
                             id_bn=dict(stats=rep_id, gamma=gam2, beta=bet2, replicas=2) if idbn else None, stats=s1, stats_replicas=2)
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref)
    assert torch.equal(y, y_ref)
    np.testing.assert_allclose(s1.sum(0).cpu().numpy(), s0.cpu().numpy(), rtol=2e-3, atol=2e-2 * np.sqrt(n))
    ref = F.conv2d(x_ref.float().cpu().permute(0, 3, 1, 2), w, None, 1, 0).permute(0, 2, 3, 1)
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * ref.abs().max().item()


# (rows shape, N of the next conv1, identity needs its own BatchNorm)
B2B_CASES = [((2, 56, 56), 64, False), ((1, 56, 56), 64, True), ((2, 56, 56), 128, False), ((3, 9, 7), 64, True), ((1, 1, 5), 128, False),
             ((5, 28, 28), 64, False),
             ((3, 28, 28), 0, False), ((2, 28, 28), 0, True), ((1, 5, 3), 0, False)]   # N = 0: the layer2 form (128 -> 512, stops at x_out)


@pytest.mark.parametrize("case", B2B_CASES)
def test_conv_b2b_equals_conv3_bn_act_conv1(case):
    """st_conv_b2b (conv3 64 -> 256 RE-computed from the narrow tensor, bn3 + identity + ReLU, next conv1 256 -> 64 | 128 in one kernel)
    against the three-kernel form st_conv1x1_wreg -> st_bn_act -> st_conv1x1_wreg: x_out and y bit for bit; the statistics-only conv3
    (y == NULL) must produce the same statistics as the storing one."""
    ops = _ops()
    shape, N, idbn = case
    B, H, W = shape
    C1, C2 = (64, 256) if N > 0 else (128, 512)
    assert ops.lib().st_conv_b2b_supported(C1, C2, N) == 1 and ops.lib().st_conv_b2b_supported(C1, C2, 256) == 0
    g = torch.Generator().manual_seed(B * H + W + N)
    raw2 = (torch.randn(B, H, W, C1, generator=g) * 1.1 + 0.2).bfloat16().cuda()
    ident = (torch.randn(B, H, W, C2, generator=g) * (1.0 if idbn else 0.7)).bfloat16().cuda()
    if not idbn:
        ident = torch.relu(ident)
    w3 = (torch.randn(C2, C1, 1, 1, generator=g) / np.sqrt(C1)).bfloat16().float()
    w3f = ops.pack_conv_weight_frag(w3.cuda(), ops.conv1x1_wreg_supported(C1, C2))
    if N > 0:
        w1 = (torch.randn(N, C2, 1, 1, generator=g) / np.sqrt(C2)).bfloat16().float()
        w1f = ops.pack_conv_weight_frag(w1.cuda(), ops.conv1x1_wreg_supported(C2, N))
    else:
        w1f = None
    g2, b2 = (torch.rand(C1, generator=g) + 0.5).cuda(), (torch.randn(C1, generator=g) * 0.3).cuda()
    g3, b3 = (torch.rand(C2, generator=g) + 0.5).cuda(), (torch.randn(C2, generator=g) * 0.3).cuda()
    gi, bi = (torch.rand(C2, generator=g) + 0.5).cuda(), (torch.randn(C2, generator=g) * 0.3).cuda()
    n = float(B * H * W)

    def stats_of(t, C):
        t2 = t.float().reshape(-1, C)
        return torch.cat([t2.sum(0), (t2 * t2).sum(0)]).contiguous()
    s2 = torch.zeros(3, 2 * C1, device="cuda"); s2[1] = stats_of(raw2, C1)
    si = torch.zeros(2, 2 * C2, device="cuda"); si[1] = stats_of(ident, C2)
    bn2 = dict(stats=s2, gamma=g2, beta=b2, count=n, replicas=3)
    # three-kernel form
    s3 = torch.zeros(4, 2 * C2, device="cuda")
    raw3 = ops.conv1x1_wreg(raw2, w3f, C2, stats=s3, stats_replicas=4, in_bn=bn2)
    s3_only = torch.zeros(4, 2 * C2, device="cuda")
    assert ops.conv1x1_wreg(raw2, w3f, C2, stats=s3_only, stats_replicas=4, in_bn=bn2, stats_only=True) is None
    torch.cuda.synchronize()
    np.testing.assert_allclose(s3_only.sum(0).cpu().numpy(), s3.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-2)
    s3r = s3.sum(0).contiguous()                            # ONE set of bn3 statistics for both forms: bit-exact comparison below
    kw = dict(res=ident)
    if idbn:
        kw.update(res_bn=dict(stats=si[1].contiguous(), gamma=gi, beta=bi))
    x_ref = ops.bn_act(raw3, g3, b3, stats=s3r, count=n, relu=True, **kw)
    # one kernel
    s3rep = torch.zeros(2, 2 * C2, device="cuda"); s3rep[1] = s3r
    sy1 = torch.zeros(4, 2 * max(N, 1), device="cuda")
    x, y = ops.conv_b2b(raw2, w3f, ident, w1f, N, dict(stats=s2, gamma=g2, beta=b2, replicas=3), dict(stats=s3rep, gamma=g3, beta=b3, replicas=2), n,
                        id_bn=dict(stats=si, gamma=gi, beta=bi, replicas=2) if idbn else None, stats=sy1 if N > 0 else None, stats_replicas=4)
    torch.cuda.synchronize()
    assert torch.equal(x, x_ref)
    if N > 0:
        sy0 = torch.zeros(2 * N, device="cuda")
        y_ref = ops.conv1x1_wreg(x_ref, w1f, N, stats=sy0)
        torch.cuda.synchronize()
        assert torch.equal(y, y_ref)
        np.testing.assert_allclose(sy1.sum(0).cpu().numpy(), sy0.cpu().numpy(), rtol=2e-3, atol=2e-2 * np.sqrt(n))
    else:
        assert y is None


# ---- st_conv_c3c1: conv3 + block end + next conv1 of the 14 x 14 (256 -> 1024 -> 256) and 28 x 28 (128 -> 512 -> 128) Bottlenecks in one kernel
C3C1_CASES = [(4, 14, 14), (1, 9, 13), (3, 7, 5), (1, 1, 3), (9, 14, 14)]      # full row blocks, ragged, fewer rows than a tile, many blocks
C3C1_GEO = [(256, 1024), (128, 512)]                                            # (C1 = N, C2)


def _c3c1_path(ops, geo):
    """The separate kernels the engine would otherwise run for this geometry: (conv3, next conv1) and their fragment `ntw`s."""
    C1, C2 = geo
    if geo == (256, 1024):
        return ops.conv1x1_astat, ops.conv1x1_kstream, ops.conv1x1_astat_supported(C1, C2), 4
    return ops.conv1x1_wreg, ops.conv1x1_wreg, ops.conv1x1_wreg_supported(C1, C2), ops.conv1x1_wreg_supported(C2, C1)


def _c3c1_data(shape, seed, geo=(256, 1024)):
    ops = _ops()
    B, H, W = shape
    C1, C2 = geo
    g = torch.Generator().manual_seed(1000 * seed + B * H + W)
    x2 = (torch.randn(B, H, W, C1, generator=g) * 1.2 + 0.2).bfloat16().cuda()
    ident = torch.relu(torch.randn(B, H, W, C2, generator=g)).bfloat16().cuda()
    w3 = (torch.randn(C2, C1, 1, 1, generator=g) / C1 ** 0.5).bfloat16().float()
    w1 = (torch.randn(C1, C2, 1, 1, generator=g) / C2 ** 0.5).bfloat16().float()
    _, _, ntw3, ntw1 = _c3c1_path(ops, geo)
    assert ntw3 == 2 and ntw1 == C1 // 64                     # the fixed tile permutations conv_c3c1_kernel indexes
    w3a = ops.pack_conv_weight_frag(w3.cuda(), ntw3)
    w1k = ops.pack_conv_weight_frag(w1.cuda(), ntw1)
    return g, x2, ident, w3, w1, w3a, w1k


@pytest.mark.parametrize("geo", C3C1_GEO)
@pytest.mark.parametrize("shape", C3C1_CASES)
def test_conv_c3c1_train_equals_astat_bn_act_kstream_bit_for_bit(shape, geo):
    """Train mode: x_out == st_bn_act(conv3(x2; bn2), bn3, res = identity) and y == conv1(x_out), bit for bit, where conv3 / conv1 are
    the kernels the engine otherwise runs (st_conv1x1_astat / st_conv1x1_kstream at 14 x 14, st_conv1x1_wreg for both at 28 x 28): same
    MFMA order, raw conv3 rounded to bf16 at the same point, same coefficients; bn3's statistics from the statistics-only pass (y == NULL)
    equal the writing pass's; y's statistics equal the separate conv1's."""
    ops = _ops()
    B, H, W = shape
    C1, C2 = geo
    conv3, conv1, _, _ = _c3c1_path(ops, geo)
    g, x2, ident, w3, w1, w3a, w1k = _c3c1_data(shape, 1, geo)
    n = float(B * H * W)
    gam2, bet2 = (torch.rand(C1, generator=g) + 0.5).cuda(), (torch.randn(C1, generator=g) * 0.5).cuda()
    gam3, bet3 = (torch.rand(C2, generator=g) + 0.5).cuda(), (torch.randn(C2, generator=g) * 0.5).cuda()
    x2f = x2.float().reshape(-1, C1)
    st2 = torch.zeros(3, 2 * C1, device="cuda"); st2[1] = torch.cat([x2f.sum(0), (x2f * x2f).sum(0)])
    bn2 = dict(stats=st2, gamma=gam2, beta=bet2, count=n, replicas=3)
    # the three-kernel path
    s3 = torch.zeros(4, 2 * C2, device="cuda")
    raw3 = conv3(x2, w3a, C2, stats=s3, stats_replicas=4, in_bn=bn2)
    x_ref = ops.bn_act(raw3, gam3, bet3, stats=s3, stats_replicas=4, count=n, relu=True, res=ident)
    s1_ref = torch.zeros(2, 2 * C1, device="cuda")
    y_ref = conv1(x_ref, w1k, C1, stats=s1_ref, stats_replicas=2)
    # statistics-only pass + the fused kernel
    s3b = torch.zeros(4, 2 * C2, device="cuda")
    assert conv3(x2, w3a, C2, stats=s3b, stats_replicas=4, in_bn=bn2, stats_only=True) is None
    torch.cuda.synchronize()
    np.testing.assert_allclose(s3b.sum(0).cpu().numpy(), s3.sum(0).cpu().numpy(), rtol=1e-5, atol=1e-3 * np.sqrt(n))
    s1 = torch.zeros(2, 2 * C1, device="cuda")
    x, y = ops.conv_c3c1(x2, w3a, ident, w1k, bn2=bn2, bn3=dict(stats=s3, gamma=gam3, beta=bet3, replicas=4), count=n, stats=s1, stats_replicas=2)
    torch.cuda.synchronize()
    assert y.shape[-1] == C1
    assert torch.equal(x, x_ref)
    assert torch.equal(y, y_ref)
    np.testing.assert_allclose(s1.sum(0).cpu().numpy(), s1_ref.sum(0).cpu().numpy(), rtol=1e-4, atol=1e-2)
    # and against fp32 arithmetic end to end (batch statistics over a handful of rows amplify the bf16 rounding of the raw tensor: skipped there)
    if n >= 64:
        a2 = F.relu(F.batch_norm(x2.float().cpu().permute(0, 3, 1, 2), None, None, gam2.cpu(), bet2.cpu(), True, 0.0, 1e-5)).bfloat16().float()
        r3 = F.conv2d(a2, w3)
        xf = F.relu(F.batch_norm(r3, None, None, gam3.cpu(), bet3.cpu(), True, 0.0, 1e-5) + ident.float().cpu().permute(0, 3, 1, 2))
        assert (x.float().cpu().permute(0, 3, 1, 2) - xf).abs().max().item() <= 2e-2 * xf.abs().max().item()
    yf = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), w1)
    assert (y.float().cpu().permute(0, 3, 1, 2) - yf).abs().max().item() <= 1.5e-2 * yf.abs().max().item()
    # the block behind a downsample conv: the identity is RAW with its own batch statistics (st_bn_act's res_bn form), bit for bit
    if n >= 2:
        idr = (torch.randn(B, H, W, C2, generator=g) * 0.7 + 0.1).bfloat16().cuda()
        idf = idr.float().reshape(-1, C2)
        std_ = torch.zeros(2, 2 * C2, device="cuda"); std_[1] = torch.cat([idf.sum(0), (idf * idf).sum(0)])
        gamd, betd = (torch.rand(C2, generator=g) + 0.5).cuda(), (torch.randn(C2, generator=g) * 0.3).cuda()
        x_ref2 = ops.bn_act(raw3, gam3, bet3, stats=s3, stats_replicas=4, count=n, relu=True, res=idr,
                            res_bn=dict(stats=std_, gamma=gamd, beta=betd, stats_replicas=2))
        y_ref2 = conv1(x_ref2, w1k, C1, stats=torch.zeros(2, 2 * C1, device="cuda"), stats_replicas=2)
        x_i, y_i = ops.conv_c3c1(x2, w3a, idr, w1k, bn2=bn2, bn3=dict(stats=s3, gamma=gam3, beta=bet3, replicas=4), count=n,
                                 stats=torch.zeros(2, 2 * C1, device="cuda"), stats_replicas=2, id_bn=dict(stats=std_, gamma=gamd, beta=betd, replicas=2))
        torch.cuda.synchronize()
        assert torch.equal(x_i, x_ref2) and torch.equal(y_i, y_ref2)
    # x2 already normalised (bn2 = None): the loader copies it
    a2d = ops.bn_act(x2, gam2, bet2, stats=st2, stats_replicas=3, count=n, relu=True)
    x_b, y_b = ops.conv_c3c1(a2d, w3a, ident, w1k, bn3=dict(stats=s3, gamma=gam3, beta=bet3, replicas=4), count=n, stats=torch.zeros(1, 2 * C1, device="cuda"), stats_replicas=1)
    assert torch.equal(x_b, x_ref) and torch.equal(y_b, y_ref)


@pytest.mark.parametrize("geo", C3C1_GEO)
@pytest.mark.parametrize("shape", C3C1_CASES[:3])
def test_conv_c3c1_eval_equals_astat_residual_then_kstream(shape, geo):
    """Eval mode (folded BatchNorms): x_out == conv3(.., scale3, shift3, relu, residual = identity), y == conv1(x_out, scale1, shift1,
    relu) of the separate kernels, bit for bit; and both against fp32."""
    ops = _ops()
    C1, C2 = geo
    conv3, conv1, _, _ = _c3c1_path(ops, geo)
    g, x2, ident, w3, w1, w3a, w1k = _c3c1_data(shape, 2, geo)
    x2 = torch.relu(x2)
    sc3, sh3 = (torch.rand(C2, generator=g) + 0.5).cuda(), (torch.randn(C2, generator=g) * 0.3).cuda()
    sc1, sh1 = (torch.rand(C1, generator=g) + 0.5).cuda(), (torch.randn(C1, generator=g) * 0.3).cuda()
    x_ref = conv3(x2, w3a, C2, scale=sc3, shift=sh3, relu=True, residual=ident)
    for relu1 in (True, False):
        y_ref = conv1(x_ref, w1k, C1, scale=sc1, shift=sh1, relu=relu1)
        x, y = ops.conv_c3c1(x2, w3a, ident, w1k, scale3=sc3, shift3=sh3, scale1=sc1, shift1=sh1, relu1=relu1)
        torch.cuda.synchronize()
        assert torch.equal(x, x_ref)
        assert torch.equal(y, y_ref)
    xf = F.relu(F.conv2d(x2.float().cpu().permute(0, 3, 1, 2), w3) * sc3.cpu()[None, :, None, None] + sh3.cpu()[None, :, None, None] + ident.float().cpu().permute(0, 3, 1, 2))
    assert (x.float().cpu().permute(0, 3, 1, 2) - xf).abs().max().item() <= 1.5e-2 * xf.abs().max().item()
    from showtell_amd import ShowTellHipError
    with pytest.raises(ShowTellHipError):
        ops.conv_c3c1(x2, w3a, ident, w1k, scale3=sc3, shift3=sh3)            # eval needs all four folded coefficient vectors


# ---- st_conv3x3_s2: the stride-2 3x3 convs (B, H, W, C = N) -------------------------------------------------------------------------
S2_CASES = [(2, 56, 56, 128), (3, 28, 28, 256), (5, 14, 14, 512), (1, 9, 7, 128), (2, 5, 5, 256), (1, 1, 1, 512), (3, 13, 15, 256)]


@pytest.mark.parametrize("case", S2_CASES)
def test_conv3x3_s2_matches_conv2d_and_igemm(case):
    """K-streaming stride-2 3x3 (odd and even maps, a single pixel): output vs F.conv2d and vs the implicit-GEMM kernel, statistics,
    producer's BatchNorm + ReLU in the loader == the separate pass bit for bit (padding stays zero after the transform), eval epilogue."""
    ops = _ops()
    B, H, W, C = case
    N = C
    ntw = ops.conv3x3_s2_supported(C, N)
    assert ntw > 0 and ops.conv3x3_s2_supported(64, 64) == 0
    g = torch.Generator().manual_seed(B * H * W + C)
    x = (torch.randn(B, H, W, C, generator=g) * 1.1 + 0.15).bfloat16()
    w = (torch.randn(N, C, 3, 3, generator=g) / np.sqrt(9 * C)).bfloat16().float()
    ref = F.conv2d(x.float().permute(0, 3, 1, 2), w, None, 2, 1).permute(0, 2, 3, 1).contiguous()
    xd = x.cuda()
    wf = ops.pack_conv_weight_frag(w.cuda(), ntw)
    st = torch.zeros(4, 2 * N, device="cuda")
    y = ops.conv3x3_s2(xd, wf, N, stats=st, stats_replicas=4)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    scale = ref.abs().max().item()
    assert (y.float().cpu() - ref).abs().max().item() <= 1.5e-2 * scale
    y0 = ops.conv_nhwc(xd, ops.pack_conv_weight(w.cuda(), torch.bfloat16, k_order=1), 3, 3, 2, 1, k_order=1)
    assert (y.float() - y0.float()).abs().max().item() <= 2.0 ** -6 * scale
    r2 = ref.reshape(-1, N)
    tot = st.sum(0).cpu().numpy()
    np.testing.assert_allclose(tot[:N], r2.sum(0).numpy(), rtol=2e-3, atol=2e-3 * scale * np.sqrt(r2.shape[0]))
    np.testing.assert_allclose(tot[N:], (r2 * r2).sum(0).numpy(), rtol=2e-3, atol=1e-4)
    # producer's BatchNorm + ReLU in the loader
    gam, bet = (torch.rand(C, generator=g) + 0.5).cuda(), (torch.randn(C, generator=g) * 0.5).cuda()
    x2 = xd.float().reshape(-1, C)
    stats = torch.cat([x2.sum(0), (x2 * x2).sum(0)]).contiguous()
    n = float(B * H * W)
    y_sep = ops.conv3x3_s2(ops.bn_act(xd, gam, bet, stats=stats, count=n, relu=True), wf, N)
    rep = torch.zeros(3, 2 * C, device="cuda"); rep[1] = stats
    y_fused = ops.conv3x3_s2(xd, wf, N, in_bn=dict(stats=rep, gamma=gam, beta=bet, count=n, replicas=3))
    assert torch.equal(y_sep, y_fused)
    # eval-mode epilogue
    sc, sh = (torch.rand(N, generator=g) + 0.5), torch.randn(N, generator=g) * 0.3
    y2 = ops.conv3x3_s2(xd, wf, N, scale=sc.cuda(), shift=sh.cuda(), relu=True)
    ref2 = F.relu(ref * sc + sh)
    assert (y2.float().cpu() - ref2).abs().max().item() <= 1.5e-2 * max(ref2.abs().max().item(), 1e-3)
