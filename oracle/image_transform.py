"""CPU restatement of the reference's per-image input transform (TEST INFRASTRUCTURE ONLY).

Reference: utils.py:84-88 -- ``tf.Compose([Resize((224, 224)), RandomHorizontalFlip(), RandomVerticalFlip(),
ToTensor(), Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))])`` applied to ``Image.open(..).convert('RGB')``
(utils.py:45-47); ``create_batch`` (utils.py:61-77) then sorts by caption length and stacks.

The arithmetic lives in third-party code absent from /root/reference:
  * torchvision 0.3.0 (README.md:24): ``Resize`` = ``PIL.Image.resize(size[::-1], Image.BILINEAR)``; the flips are
    ``Image.transpose(FLIP_LEFT_RIGHT / FLIP_TOP_BOTTOM)``; ``ToTensor`` = HWC uint8 -> CHW float32 ``.div(255)``;
    ``Normalize`` = ``(t - mean) / std`` per channel in float32.
  * Pillow (``src/libImaging/Resample.c``): ``precompute_coeffs`` (double arithmetic), ``normalize_coeffs_8bpc``
    (22-bit fixed point), ``ImagingResampleHorizontal_8bpc`` then ``ImagingResampleVertical_8bpc`` with a uint8
    intermediate image.  The published algorithm is restated below.

Pinning: ``tests/golden/make_image_golden.py`` ran Pillow 12.2.0 (installed in the authoring container) and torch CPU
ops on deterministic images and stored the outputs in ``tests/golden/image_transform.npz``;
``tests/test_image_transform.py`` checks this restatement against them bit for bit.

Only ``tests/`` may import this file.  The product path (``show-tell_amd/data.py`` + ``csrc/preprocess.hip``)
never does.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2            # Resample.c: fixed-point scale of the 8-bit-per-channel paths
MEAN = (0.485, 0.456, 0.406)           # utils.py:88
STD = (0.229, 0.224, 0.225)


def synthetic_image(h, w, seed):
    """Deterministic uint8 RGB test image (integer hash + smooth ramp): no RNG-version dependence."""
    i = np.arange(h, dtype=np.uint64)[:, None, None]
    j = np.arange(w, dtype=np.uint64)[None, :, None]
    c = np.arange(3, dtype=np.uint64)[None, None, :]
    hsh = (i * np.uint64(73856093)) ^ (j * np.uint64(19349663)) ^ ((c + np.uint64(seed)) * np.uint64(83492791))
    hsh = (hsh * np.uint64(2654435761)) >> np.uint64(13)
    noise = (hsh & np.uint64(63)).astype(np.int64)
    ramp = (i.astype(np.int64) * 3 + j.astype(np.int64) * 5 + c.astype(np.int64) * 40 + seed * 17) % 192
    return (ramp + noise).astype(np.uint8)


def bilinear_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the triangle filter (support 1.0).
    Returns (xmin[out], count[out], k[out][ksize] int32)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size       # (double)(in1 - in0) / outSize, box = full image
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xmin = np.zeros(out_size, np.int64)
    cnt = np.zeros(out_size, np.int64)
    kk = np.zeros((out_size, ksize), np.int64)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        lo = int(center - support + 0.5)              # C (int) cast: truncation toward zero
        lo = max(lo, 0)
        hi = int(center + support + 0.5)
        hi = min(hi, in_size)
        n = hi - lo
        w = []
        ww = 0.0
        for x in range(n):
            a = (x + lo - center + 0.5) * ss
            a = -a if a < 0.0 else a
            v = 1.0 - a if a < 1.0 else 0.0
            w.append(v)
            ww += v
        for x in range(n):
            v = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + v * (1 << PRECISION_BITS)) if v < 0 else int(0.5 + v * (1 << PRECISION_BITS))
        xmin[xx], cnt[xx] = lo, n
    return xmin, cnt, kk


def _resample_axis0(img, out_size):
    """One 8bpc pass along axis 0 of a uint8 array: out[o] = clip8((2^21 + sum_x img[xmin+x] * k[o][x]) >> 22)."""
    xmin, cnt, kk = bilinear_coeffs(img.shape[0], out_size)
    out = np.empty((out_size,) + img.shape[1:], np.uint8)
    src = img.astype(np.int64)
    for o in range(out_size):
        acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
        for x in range(cnt[o]):
            acc += src[xmin[o] + x] * kk[o, x]
        out[o] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return out


def resize_bilinear_u8(img, out_h, out_w):
    """PIL.Image.resize((out_w, out_h), BILINEAR) of an HWC uint8 image: horizontal pass, uint8 intermediate, vertical
    pass (ImagingResample; a pass whose size does not change is skipped there -- its coefficients are (1, 0), the identity)."""
    tmp = img
    if img.shape[1] != out_w:
        tmp = np.swapaxes(_resample_axis0(np.swapaxes(img, 0, 1), out_w), 0, 1)
    if img.shape[0] != out_h:
        tmp = _resample_axis0(tmp, out_h)
    return np.ascontiguousarray(tmp)


def normalize_lut():
    """[3][256] float32: ToTensor's ``v / 255`` followed by Normalize's ``(t - mean) / std``, all in float32."""
    v = np.arange(256, dtype=np.float32) / np.float32(255)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(MEAN, STD)]).astype(np.float32)


def transform(img, hflip=False, vflip=False, size=(224, 224)):
    """utils.py:84-88 on one HWC uint8 RGB image -> (3, H, W) float32 (the coin flips are the caller's)."""
    r = resize_bilinear_u8(img, size[0], size[1])
    if hflip:
        r = r[:, ::-1]
    if vflip:
        r = r[::-1]
    lut = normalize_lut()
    return np.stack([lut[c][r[:, :, c]] for c in range(3)])


def create_batch(data):
    """utils.py:61-77: sort by caption length (descending, stable), stack images, zero-pad captions."""
    data = sorted(data, key=lambda x: len(x[2]), reverse=True)
    paths, images, captions = zip(*data)
    lens = [len(c) for c in captions]
    target = np.zeros((len(captions), max(lens)), np.int64)
    for i, c in enumerate(captions):
        target[i, :lens[i]] = np.asarray(c[:lens[i]])
    return paths, np.stack(images, 0), target, lens
