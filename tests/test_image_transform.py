"""Input transform (utils.py:61-88): oracle vs the Pillow / torch golden outputs on CPU, the HIP
kernels vs both on the GPU.  Bit-exact everywhere (byte work; the float result is a table lookup)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import image_transform as O  # noqa: E402

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "image_transform.npz"))
CASES = [tuple(int(v) for v in row) for row in GOLD["cases"]]


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


@pytest.mark.parametrize("n", [n for n, c in enumerate(CASES) if c[0] * c[1] <= 700 * 700])
def test_oracle_resize_matches_pillow_golden(n):
    h, w, seed, hf, vf = CASES[n]
    r = O.resize_bilinear_u8(O.synthetic_image(h, w, seed), 224, 224)
    r = r[:, ::-1] if hf else r
    r = r[::-1] if vf else r
    assert np.array_equal(r, GOLD[f"u8_{n}"])


def test_oracle_normalize_matches_torch_golden():
    assert np.array_equal(_bits(O.normalize_lut()), _bits(GOLD["lut"]))
    f = O.transform(O.synthetic_image(640, 480, 2), hflip=True, vflip=False)
    assert np.array_equal(_bits(f), _bits(GOLD["f32_1"]))


def test_host_table_and_create_batch():
    from showtell_amd.data import create_batch, normalize_table
    assert np.array_equal(_bits(normalize_table()), _bits(GOLD["lut"]))
    g = torch.Generator().manual_seed(3)
    lens = [5, 9, 5, 7, 9, 3]
    data = [(f"img{i}.jpg", torch.randn(3, 8, 8, generator=g), torch.randint(1, 50, (n,), generator=g)) for i, n in enumerate(lens)]
    paths, images, target, cl = create_batch(list(data))
    rp, ri, rt, rl = O.create_batch([(p, im.numpy(), c.numpy()) for p, im, c in data])
    assert paths == rp and cl == rl == [9, 9, 7, 5, 5, 3]          # stable: img1 before img4, img0 before img2
    assert paths[:2] == ("img1.jpg", "img4.jpg") and paths[3:5] == ("img0.jpg", "img2.jpg")
    assert target.dtype == torch.long and np.array_equal(target.numpy(), rt) and np.array_equal(images.numpy(), ri)


# ---------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_device_transform_matches_pillow_golden_ragged_batch():
    """All golden cases as ONE ragged minibatch (1x1 .. 1200x1600, mixed flips) through st_image_transform."""
    from showtell_amd.data import DeviceTransform
    imgs = [O.synthetic_image(h, w, s) for h, w, s, _, _ in CASES]
    out, u8 = DeviceTransform()(imgs, hflip=[c[3] for c in CASES], vflip=[c[4] for c in CASES], return_u8=True)
    torch.cuda.synchronize()
    u8, out = u8.cpu().numpy(), out.cpu().numpy()
    for n in range(len(CASES)):
        assert np.array_equal(u8[n], GOLD[f"u8_{n}"]), f"case {n} {CASES[n]}"
    assert out.shape == (len(CASES), 3, 224, 224)
    assert np.array_equal(_bits(out[1]), _bits(GOLD["f32_1"]))
    lut = GOLD["lut"]
    for n in range(len(CASES)):                                     # ToTensor + Normalize of the golden pixels
        want = np.stack([lut[c][GOLD[f"u8_{n}"][:, :, c]] for c in range(3)])
        assert np.array_equal(_bits(out[n]), _bits(want))


@pytest.mark.gpu
@pytest.mark.parametrize("size", [(224, 224), (96, 128), (299, 299)])
def test_device_transform_matches_oracle_other_sizes(size):
    from showtell_amd.data import DeviceTransform
    shapes = [(97, 311), (311, 97), (500, 375), (size[0], size[1]), (size[0] * 2, size[1] * 2), (7, 5), (640, 640), (481, 641)]
    imgs = [O.synthetic_image(h, w, 30 + i) for i, (h, w) in enumerate(shapes)]
    hf = [i % 2 == 1 for i in range(len(imgs))]
    vf = [i % 3 == 0 for i in range(len(imgs))]
    out = DeviceTransform(size=size)(imgs, hflip=hf, vflip=vf).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(_bits(out[i]), _bits(O.transform(im, hf[i], vf[i], size))), shapes[i]


@pytest.mark.gpu
def test_device_transform_full_batch_feeds_the_encoder():
    """A COCO-shaped minibatch of 128 images through create_batch(transform=...) into ResNet.forward; random flips."""
    import random
    from showtell_amd.cnn import ResNet
    from showtell_amd.data import DeviceTransform, create_batch
    random.seed(5)
    shapes = [(480, 640), (640, 480), (427, 640), (375, 500)]
    base = [O.synthetic_image(h, w, 50 + i) for i, (h, w) in enumerate(shapes)]
    data = [(f"{i}.jpg", base[i % 4], torch.randint(1, 90, (6 + i % 11,))) for i in range(128)]
    paths, images, cap, lens = create_batch(data, DeviceTransform())
    assert images.shape == (128, 3, 224, 224) and images.is_cuda and lens == sorted(lens, reverse=True)
    assert cap.shape == (128, 16) and int((cap != 0).sum()) == sum(lens)
    # every image is one of the 4 sources under one of 4 flip states
    i0 = paths.index("0.jpg")
    cands = [O.transform(base[0], h, v) for h in (False, True) for v in (False, True)]
    got = images[i0].cpu().numpy()
    assert any(np.array_equal(_bits(got), _bits(c)) for c in cands)
    torch.manual_seed(0)
    cnn = ResNet(18, 64, dtype=torch.bfloat16).cuda().eval()
    feat = cnn(images)
    assert feat.shape == (128, 64) and bool(torch.isfinite(feat.float()).all())


@pytest.mark.gpu
def test_device_transform_packed_buffer_equals_list_path():
    from showtell_amd.data import DeviceTransform
    shapes = [(120, 90), (64, 333), (224, 224), (300, 200)]
    imgs = [O.synthetic_image(h, w, 80 + i) for i, (h, w) in enumerate(shapes)]
    hf, vf = [True, False, False, True], [False, False, True, True]
    tf = DeviceTransform()
    want = tf(imgs, hflip=hf, vflip=vf)
    flat = torch.from_numpy(np.concatenate([a.reshape(-1) for a in imgs]))
    hs, ws = [s_[0] for s_ in shapes], [s_[1] for s_ in shapes]
    for buf in (flat.pin_memory(), flat.cuda()):
        assert torch.equal(tf.packed(buf, hs, ws, hf, vf), want)
    with pytest.raises(ValueError):
        tf.packed(flat[:-1].cuda(), hs, ws, hf, vf)


@pytest.mark.gpu
def test_device_transform_errors():
    from showtell_amd import ShowTellHipError
    from showtell_amd.data import DeviceTransform
    with pytest.raises(ValueError):
        DeviceTransform()([np.zeros((4, 4), np.uint8)])
    with pytest.raises(ValueError):
        DeviceTransform()([])
    with pytest.raises(ShowTellHipError):                            # 5000 / 224 needs 47 taps (limit 40)
        DeviceTransform()([np.zeros((10, 5000, 3), np.uint8)])
    with pytest.raises(ShowTellHipError):
        DeviceTransform(size=(512, 512))([np.zeros((10, 10, 3), np.uint8)])
