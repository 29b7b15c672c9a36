"""Writes tests/golden/image_transform.npz (authoring container only: needs Pillow).

    python tests/golden/make_image_golden.py

The fixture pins the input transform of utils.py:84-88 to the libraries that execute it for the reference: Pillow's
``Image.resize(.., BILINEAR)`` / ``Image.transpose`` and torch CPU float32 ops for ToTensor + Normalize (torchvision
itself is not installed; its two-line definitions of those transforms are restated here).  Inputs are regenerated
from ``oracle.image_transform.synthetic_image`` (pure integer arithmetic), so only outputs are stored: the resized
uint8 images, the normalisation table, and one full float32 result.
"""
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.image_transform import MEAN, STD, synthetic_image  # noqa: E402

# (h, w, seed, hflip, vflip): COCO's two common shapes, odd sizes, no-op axes, up-scaling, extreme down-scaling, tiny
CASES = [(480, 640, 1, 0, 0), (640, 480, 2, 1, 0), (427, 640, 3, 0, 1), (333, 500, 4, 1, 1), (224, 224, 5, 0, 0),
         (224, 500, 6, 1, 0), (375, 224, 7, 0, 1), (100, 150, 8, 1, 1), (1200, 1600, 9, 0, 0), (51, 72, 10, 1, 0),
         (1, 1, 11, 0, 0), (2, 3, 12, 1, 1), (225, 223, 13, 0, 0)]


def reference_transform(img, hflip, vflip):
    pil = Image.fromarray(img, "RGB").resize((224, 224), Image.BILINEAR)       # tf.Resize((224, 224))
    if hflip:
        pil = pil.transpose(Image.FLIP_LEFT_RIGHT)                             # tf.RandomHorizontalFlip, coin = heads
    if vflip:
        pil = pil.transpose(Image.FLIP_TOP_BOTTOM)                             # tf.RandomVerticalFlip
    u8 = np.asarray(pil).copy()
    t = torch.from_numpy(u8).permute(2, 0, 1).contiguous().float().div(255)   # tf.ToTensor
    mean = torch.tensor(MEAN, dtype=torch.float32)[:, None, None]
    std = torch.tensor(STD, dtype=torch.float32)[:, None, None]
    return u8, t.sub_(mean).div_(std).numpy()                                  # tf.Normalize


if __name__ == "__main__":
    out = {"cases": np.asarray(CASES, np.int64)}
    for n, (h, w, seed, hf, vf) in enumerate(CASES):
        u8, f32 = reference_transform(synthetic_image(h, w, seed), hf, vf)
        out[f"u8_{n}"] = u8
        if n == 1:
            out["f32_1"] = f32
    v = torch.arange(256, dtype=torch.float32).div(255)
    mean, std = torch.tensor(MEAN, dtype=torch.float32)[:, None], torch.tensor(STD, dtype=torch.float32)[:, None]
    out["lut"] = v[None, :].repeat(3, 1).sub_(mean).div_(std).numpy()
    import PIL
    out["pillow_version"] = np.asarray(PIL.__version__)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "image_transform.npz"), **out)
    print("wrote", len(CASES), "cases")
