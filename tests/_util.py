"""Shared helpers for the test-suite (fixtures -> oracle params)."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_fixture(name):
    d = np.load(os.path.join(GOLDEN, name))
    params = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("p/")}
    grads = {k[2:]: torch.from_numpy(d[k]) for k in d.files if k.startswith("g/")}
    rest = {k: d[k] for k in d.files if not (k.startswith("p/") or k.startswith("g/"))}
    return params, grads, rest


def req_grad(params):
    return {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in params.items()}
