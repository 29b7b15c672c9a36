"""GPU parity of the ResNet backbone engine (st_resnet_forward) against the oracle.

The oracle backbone is torch CPU fp32 (F.conv2d / F.batch_norm), "parity unpinned" at the
torchvision boundary (see oracle/restatement.py header).  fp32 kernels are compared at
1e-3 of the output scale (104 layers of re-associated fp32 sums); bf16 kernels store
every activation in bf16 (2^-8 relative rounding per layer), compared at 6e-2.
"""
import pytest
import torch
import torch.nn.functional as F

from oracle import restatement as R

pytestmark = pytest.mark.gpu
TOL = {torch.float32: 1e-3, torch.bfloat16: 6e-2}


def _rel(got, ref):
    got, ref = got.float().cpu(), ref.float()
    return ((got - ref).abs().max() / (ref.abs().max() + 1e-6)).item()


def _make(version, dtype, attn=False, seed=1, damp=True):
    from showtell_amd.cnn import ResNet
    from showtell_amd.cnn_attn import ResNet as ResNetAttn
    params = R.init_encoder_params(version, 64, seed=seed)
    g = torch.Generator().manual_seed(seed + 100)
    for k in params:   # non-trivial BN affine / running stats so eval mode is exercised properly
        if k.endswith("running_mean"):
            params[k] = torch.randn(params[k].shape, generator=g) * 0.1
        elif k.endswith("running_var"):
            params[k] = torch.rand(params[k].shape, generator=g) + 0.5
        elif ".bn" in k or k.startswith("model.1.") or "downsample.1" in k:
            if k.endswith(".weight"):
                params[k] = torch.rand(params[k].shape, generator=g) * 0.5 + 0.75
                last_bn = ".bn3." if version >= 50 else ".bn2."
                if last_bn in k and damp:
                    # Damp the residual branches as in a trained network.  With Kaiming-random branches at
                    # full scale every block amplifies ANY perturbation ~1.25x (measured with an emulation
                    # of the storage rounding: test_bf16_amplification_is_a_weight_property below measures and prints it):
                    # 33 blocks turn a 2^-9 rounding into O(1), which
                    # says nothing about the kernels.  Per-layer parity is covered by test_gpu_encoder_kernels.
                    params[k] = torch.rand(params[k].shape, generator=g) * 0.2 + 0.1
            elif k.endswith(".bias"):
                params[k] = torch.randn(params[k].shape, generator=g) * 0.1
    m = (ResNetAttn if attn else ResNet)(version, 64, dtype=dtype)
    m.load_state_dict(params)
    return m.cuda(), {k: v.clone() for k, v in params.items()}


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("version,size,B", [(18, 64, 3), (50, 64, 2), (101, 224, 2)])
@pytest.mark.parametrize("train", [False, True])
def test_backbone_matches_oracle(version, size, B, train, dtype):
    if train and size < 128:
        # batch-statistics BN over a 2x2 map of 2 images (8 samples/channel) is ill-conditioned:
        # it amplifies the bf16 storage rounding.  Give the last stage 4x4x4 = 64 samples.
        size, B = 128, 4
    m, params = _make(version, dtype, damp=(dtype == torch.bfloat16))   # fp32 runs at full scale; bf16 storage: see _make
    m.train(train)
    x = torch.randn(B, 3, size, size, generator=torch.Generator().manual_seed(5))
    ref = R.backbone_forward(params, x, version, train=train, avgpool=True).flatten(1)
    got = m.backbone_features(x.cuda())
    torch.cuda.synchronize()
    assert got.shape == ref.shape
    assert _rel(got, ref) < TOL[dtype]
    if train:  # running buffers updated like nn.BatchNorm2d (momentum 0.1, unbiased var) on every layer
        sd = m.state_dict()
        for k in ("model.1.running_mean", "model.1.running_var", "model.7.1.bn2.running_var", "model.5.0.downsample.1.running_mean"):
            assert _rel(sd[k], params[k]) < (1e-3 if dtype == torch.float32 else 5e-2), k
        assert int(sd["model.1.num_batches_tracked"]) == 1 and int(sd["model.7.0.bn1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_attention_encoder_layout(dtype):
    m, params = _make(50, dtype, attn=True, damp=(dtype == torch.bfloat16))
    m.eval()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(6))
    ref = R.encoder_attn_forward(params, x, 50, train=False)
    got = m(x.cuda())
    assert got.shape == (2, 2048, 49) and got.dtype == torch.float32    # cnn_attn.py:49
    assert _rel(got, ref) < TOL[dtype]


def test_bad_version_and_cpu_input():
    from showtell_amd import ShowTellHipError
    from showtell_amd.cnn import ResNet
    with pytest.raises(ValueError):
        ResNet(42)                                   # cnn.py:33
    m = ResNet(18, 32)
    with pytest.raises(ShowTellHipError):
        m.backbone_features(torch.zeros(1, 3, 64, 64))   # CPU tensor: no fallback


@pytest.mark.parametrize("train", [False, True])
def test_resnet101_fp32_undamped_matches_oracle(train):
    """fp32 needs no damping: the full-scale random ResNet-101 @224 (every BN gain in [0.75, 1.25], residual branches
    included) against the oracle at 1e-3, train and eval mode."""
    m, params = _make(101, torch.float32, damp=False)
    m.train(train)
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(15))
    ref = R.backbone_forward(params, x, 101, train=train, avgpool=True).flatten(1)
    got = m.backbone_features(x.cuda())
    err = _rel(got, ref)
    print(f"fp32 undamped ResNet-101 train={train}: rel err {err:.3e}")
    assert err < 1e-3
    # per-element check too (max|diff|/max|ref| is loose for small channels): 99.9 % of the features within 1e-3 of their own scale
    g, r = got.float().cpu(), ref.float()
    frac = ((g - r).abs() <= 1e-3 * (r.abs() + 0.05 * r.abs().max())).float().mean().item()
    assert frac > 0.999, frac


def test_attention_encoder_resnet101_train_mode():
    """cnn_attn.ResNet (cnn_attn.py:44-52) as main_attn.py:112 runs it: ResNet-101, train-mode BatchNorm, (B, 2048, 49)."""
    m, params = _make(101, torch.float32, attn=True, damp=False)
    m.train()
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(16))
    ref = R.encoder_attn_forward(params, x, 101, train=True)
    got = m(x.cuda())
    assert got.shape == (2, 2048, 49) and got.dtype == torch.float32
    # un-pooled map, undamped weights, 98 samples per channel: the re-associated fp32 sums (atomics included) of 104 layers
    # are amplified ~1.25x per block (test_bf16_amplification_is_a_weight_property): measured 8e-4 of the scale in the L2
    # sense, the maximum over 200 k elements a few times that; the pooled features of the same network pass 1e-3 above
    assert ((got.float().cpu() - ref).norm() / ref.norm()).item() < 2e-3
    assert _rel(got, ref) < 1e-2
    mb, pb = _make(101, torch.bfloat16, attn=True)          # bf16 storage: damped residual gains (see _make)
    mb.train()
    refb = R.encoder_attn_forward(pb, x, 101, train=True)
    gotb = mb(x.cuda()).float().cpu()
    # un-pooled features, only 98 samples per channel in the last stage's batch statistics: the 104 bf16 storage roundings
    # come out at 8.5 % (L2) on this network -- measured identically with the round-1 implicit-GEMM kernels
    # (ST_CONV_IMG=0: 8.53e-2) and the image-resident / register-filter kernels (8.59e-2), i.e. a property of bf16 storage at
    # B = 2, not of a kernel; the pooled features of the same forward pass TOL[bf16] in test_backbone_matches_oracle
    l2, mx = ((gotb - refb).norm() / refb.norm()).item(), _rel(gotb, refb)
    print(f"bf16 un-pooled ResNet-101 train: L2 rel {l2:.3e}, max rel {mx:.3e}")
    assert l2 < 0.12 and mx < 0.25


def test_bf16_amplification_is_a_weight_property():
    """Why the bf16 end-to-end tests damp the last BN gain of every block: measured here, not argued.  The SAME fp32
    kernels run on an input perturbed by one bf16 rounding (2^-9 relative); with full-scale random residual branches the
    perturbation grows by orders of magnitude through 33 blocks, with damped branches it does not.  The kernels are
    identical in both runs, so the growth is a property of the random weights."""
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(17))
    xp = x.bfloat16().float()                        # one storage rounding of the input
    out = {}
    for damp in (False, True):
        m, _ = _make(101, torch.float32, damp=damp)
        m.eval()
        a, b = m.backbone_features(x.cuda()), m.backbone_features(xp.cuda())
        rel_in = ((x - xp).norm() / x.norm()).item()
        out[damp] = ((a - b).norm() / a.norm()).item() / rel_in
    print(f"relative (L2) growth of a bf16 input rounding through ResNet-101 (eval, fp32 kernels): "
          f"undamped x{out[False]:.2f}, damped x{out[True]:.2f}")
    assert out[True] < out[False]


@pytest.mark.parametrize("train", [False, True])
def test_resnet101_bf16_block_by_block_undamped(train):
    """bf16 parity that no weight property can blur: ResNet-101 @224 with UNDAMPED weights (every BN gain in [0.75, 1.25], the
    residual branches at full scale -- the bench configuration's regime), walked block by block.  The engine hands out every
    residual block's output as it stored it (st_resnet_set_taps); the oracle's block k (restatement.block_forward = torchvision
    Bottleneck.forward in fp32) is applied to the HIP path's OWN bf16 output of block k-1, so each comparison sees one block's
    three convolutions + BatchNorms and nothing of the ~1.25x-per-block amplification of earlier roundings.  All 33 blocks, train
    and eval BatchNorm; block 0 is compared from the image (stem + pool + block 0).  Bound: 2e-2 of the block output's scale
    (max norm) and 1e-2 in the L2 sense: three bf16 storage roundings (2^-9 each) + bf16 filters."""
    m, params = _make(101, torch.bfloat16, damp=False)
    m.train(train)
    B = 4 if train else 2           # train: 4 x 7 x 7 = 196 samples per channel in layer4's batch statistics
    x = torch.randn(B, 3, 224, 224, generator=torch.Generator().manual_seed(21))
    outs, _ = m._bb.block_outputs(x.cuda(), train)
    assert len(outs) == 33
    p = {k: v.clone() for k, v in params.items()}
    taps = {}
    with torch.no_grad():
        # block 0 from the image: the oracle's stem + pool + block 0 (the fused stem keeps no normalised pool output to start from)
        xin = F.max_pool2d(F.relu(R._bn2d(p, F.conv2d(x, p["model.0.weight"], None, 2, 3), "model.1", train)), 3, 2, 1)
        worst = (0.0, 0.0, -1)
        errs = []
        k = 0
        for li, nb in enumerate(R.RESNET_SPECS[101][1]):
            for bi in range(nb):
                ref = R.block_forward(p, xin, 101, li, bi, train)                     # (B, C, h, w) fp32
                got = outs[k].float().cpu().permute(0, 3, 1, 2)
                assert got.shape == ref.shape, (k, got.shape, ref.shape)
                mx = ((got - ref).abs().max() / ref.abs().max()).item()
                l2 = ((got - ref).norm() / ref.norm()).item()
                if mx > worst[0]:
                    worst = (mx, l2, k)
                errs.append((k, f"layer{li + 1}.{bi}", mx, l2))
                xin = got.contiguous()                                                # the HIP path's own output feeds the next oracle block
                k += 1
    print(f"bf16 block-by-block ResNet-101 train={train}: worst block {worst[2]}: max-rel {worst[0]:.3e}, L2 {worst[1]:.3e}")
    print("  " + "  ".join(f"{n}:{mx:.1e}/{l2:.1e}" for _, n, mx, l2 in errs))
    for k, n, mx, l2 in errs:
        # block 0 is compared from the fp32 IMAGE: bf16 image, 7x7 stem, pool and the four convolutions of layer1.0 -- twice the
        # roundings of any other block
        lim_mx, lim_l2 = (3e-2, 2e-2) if k == 0 else (2e-2, 1e-2)
        assert mx < lim_mx and l2 < lim_l2, f"block {k} ({n}, train={train}): max {mx:.3e}, L2 {l2:.3e}"
