"""Device-side input pipeline: the reference's per-image transform and collate on the MI355X.

The reference decodes each image with PIL and runs, per image and on CPU workers (utils.py:45-47, 84-88)::

    tf.Compose([tf.Resize((224, 224)), tf.RandomHorizontalFlip(), tf.RandomVerticalFlip(), tf.ToTensor(),
                tf.Normalize((0.485, 0.456, 0.406), (0.229, 0.224, 0.225))])

then ``create_batch`` (utils.py:61-77) sorts the minibatch by caption length and stacks.  At ~20 k images/s per GPU
the PIL resize of that CPU path is the bottleneck long before the kernels are, so here the decoded uint8 pixels
(ragged sizes) are copied to HBM once, back to back, and ``st_image_transform`` (csrc/preprocess.hip) does the rest
in two launches: Pillow's 8-bit BILINEAR resample bit for bit, the flips, ToTensor and Normalize
(190 us per 128 COCO-sized images, tools/time_transform.py).
There is no CPU fallback.
"""
import ctypes as C
import random

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

MEAN = (0.485, 0.456, 0.406)   # utils.py:88
STD = (0.229, 0.224, 0.225)


def normalize_table(mean=MEAN, std=STD):
    """[3][256] float32: ToTensor's ``v / 255`` then Normalize's ``(t - mean) / std``, each step rounded to float32
    as the reference's float32 tensors are."""
    v = np.arange(256, dtype=np.float32) / np.float32(255)
    return np.stack([(v - np.float32(m)) / np.float32(s) for m, s in zip(mean, std)]).astype(np.float32)


class DeviceTransform:
    """``data_transform`` of utils.py:84-88 for a whole minibatch of decoded RGB images.

    ``__call__(images, hflip=None, vflip=None)``: ``images`` is a sequence of HWC uint8 arrays (numpy, CPU torch
    tensors or PIL RGB images) of any sizes; returns the ``(B, 3, H, W)`` float32 CUDA tensor that
    ``torch.stack`` of the reference's per-image results would hold.  ``hflip`` / ``vflip``: per-image booleans; by
    default each is drawn as torchvision 0.3.0 does (``random.random() < p``, horizontal then vertical, image by image).
    """

    def __init__(self, size=(224, 224), mean=MEAN, std=STD, hflip_p=0.5, vflip_p=0.5, device="cuda"):
        self.size = (int(size[0]), int(size[1]))
        self.hflip_p, self.vflip_p = hflip_p, vflip_p
        self.device = torch.device(device)
        self._table = normalize_table(mean, std)
        self._lut = None

    def __call__(self, images, hflip=None, vflip=None, return_u8=False):
        if not torch.cuda.is_available():
            raise _lib.ShowTellHipError("DeviceTransform needs the GPU (the HIP path has no CPU fallback)")
        arrs = []
        for im in images:
            a = im.numpy() if isinstance(im, torch.Tensor) else np.asarray(im)
            if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
                raise ValueError("DeviceTransform expects HWC uint8 RGB images, got %s %s" % (a.dtype, a.shape))
            arrs.append(np.ascontiguousarray(a))
        if not arrs:
            raise ValueError("DeviceTransform: empty batch")
        host = torch.empty(sum(a.size for a in arrs), dtype=torch.uint8, pin_memory=True)
        hv, o = host.numpy(), 0
        for a in arrs:
            hv[o:o + a.size] = a.reshape(-1)
            o += a.size
        return self.packed(host, [a.shape[0] for a in arrs], [a.shape[1] for a in arrs], hflip, vflip, return_u8)

    def packed(self, pixels, heights, widths, hflip=None, vflip=None, return_u8=False):
        """The same for pixels that are already back to back in ONE uint8 tensor (image b = ``heights[b] * widths[b] * 3``
        bytes, HWC): a pinned host buffer that the decode workers filled in place (one asynchronous copy, no staging pass
        on this thread) or a CUDA tensor."""
        if not torch.cuda.is_available():
            raise _lib.ShowTellHipError("DeviceTransform needs the GPU (the HIP path has no CPU fallback)")
        B = len(heights)
        if B == 0 or len(widths) != B:
            raise ValueError("DeviceTransform: empty batch or heights / widths mismatch")
        if hflip is None:
            coins = [(random.random() < self.hflip_p, random.random() < self.vflip_p) for _ in range(B)]
            hflip, vflip = [c[0] for c in coins], [c[1] for c in coins]
        elif vflip is None:
            vflip = [False] * B
        meta = np.empty((3, B), np.int32)
        meta[0], meta[1] = heights, widths
        meta[2] = [int(bool(h)) | (int(bool(v)) << 1) for h, v in zip(hflip, vflip)]
        if meta[0].min() < 1 or meta[1].min() < 1:
            raise ValueError("DeviceTransform: empty image")
        sizes = meta[0].astype(np.int64) * meta[1] * 3
        offs = np.concatenate([[0], np.cumsum(sizes)[:-1]]).astype(np.int64)
        if pixels.dtype != torch.uint8 or pixels.dim() != 1 or pixels.numel() < int(sizes.sum()):
            raise ValueError("DeviceTransform: pixels must be a 1-D uint8 tensor holding all %d bytes" % int(sizes.sum()))
        dev = self.device
        with torch.cuda.device(dev):
            src = pixels.to(dev, non_blocking=True)
            d_off = torch.from_numpy(offs).pin_memory().to(dev, non_blocking=True)
            d_meta = torch.from_numpy(meta).pin_memory().to(dev, non_blocking=True)
            if self._lut is None or self._lut.device != src.device:
                self._lut = torch.from_numpy(self._table).to(dev)
            oh, ow = self.size
            max_h, max_w = int(meta[0].max()), int(meta[1].max())
            tmp = torch.empty(B * max_h * ow * 3, dtype=torch.uint8, device=dev)
            out = torch.empty(B, 3, oh, ow, dtype=torch.float32, device=dev)
            u8 = torch.empty(B, oh, ow, 3, dtype=torch.uint8, device=dev) if return_u8 else None
            p = lambda t: None if t is None else C.c_void_p(t.data_ptr())   # noqa: E731
            d = _lib.ImageBatchDesc(p(src), src.numel(), p(d_off), p(d_meta[0]), p(d_meta[1]), p(d_meta[2]), B, max_h, max_w, oh, ow,
                                    p(self._lut), p(tmp), p(out), p(u8))
            check(lib().st_image_transform(C.byref(d), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "st_image_transform")
        return (out, u8) if return_u8 else out


def create_batch(data, transform=None):
    '''Function to create batches from images and the corresponding real captions (utils.py:61-77).

    ``data``: list of ``(image_path, image, caption)``.  With ``transform=None`` the images are already ``(3, H, W)``
    tensors and are stacked like the reference does; with a ``DeviceTransform`` they are the decoded uint8 pixels and the
    whole (sorted) minibatch is transformed on the GPU.  Returns ``(image_paths, images, target_captions, caption_len)``.
    '''
    order = sorted(range(len(data)), key=lambda i: len(data[i][2]), reverse=True)   # stable, like list.sort (utils.py:66)
    image_paths = tuple(data[i][0] for i in order)
    pixels = [data[i][1] for i in order]
    captions = [torch.as_tensor(data[i][2]) for i in order]
    caption_len = [int(c.numel()) for c in captions]
    images = torch.stack(pixels, 0) if transform is None else transform(pixels)
    target_captions = torch.zeros(len(order), max(caption_len), dtype=torch.long)   # pad id 0 = <pad>
    for row, c in zip(target_captions, captions):
        row[:c.numel()] = c
    return image_paths, images, target_captions, caption_len
