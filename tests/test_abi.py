"""The C-ABI library loads without a GPU and exports every symbol include/showtell_hip.h declares."""
import ctypes
import os
import re

from tests._util import ROOT


def _header_symbols():
    txt = open(os.path.join(ROOT, "include", "showtell_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"#ifdef ST_EXPERIMENTAL.*?#endif", "", txt, flags=re.S)   # `make EXPERIMENTAL=1` entry points are not in the product build
    return sorted(set(re.findall(r"\b(st_[a-z0-9_]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from showtell_amd import _lib
    L = _lib.lib()                      # raises ShowTellHipError when the .so is missing: no fallback
    assert L.st_version() >= 1
    syms = _header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(L, s), f"{s} declared in showtell_hip.h but not exported"
    # every bound signature corresponds to a declared symbol
    assert set(_lib.declared_symbols()) <= set(syms), set(_lib.declared_symbols()) - set(syms)


def test_struct_sizes_match_the_header():
    """ctypes mirrors of the C structs must have the C compiler's layout (checked with gcc)."""
    import subprocess, tempfile
    from showtell_amd import _lib
    src = ('#include <stdio.h>\n#include "showtell_hip.h"\nint main(){printf("%zu %zu %zu %zu %zu\\n", sizeof(st_conv_desc), '
           'sizeof(st_bn_act_desc), sizeof(st_rnn_params), sizeof(st_rnn_grads), sizeof(st_packed_seq));return 0;}\n')
    with tempfile.TemporaryDirectory() as td:
        c = os.path.join(td, "s.c")
        open(c, "w").write(src)
        exe = os.path.join(td, "s")
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), c, "-o", exe])
        sizes = list(map(int, subprocess.check_output([exe]).split()))
    assert sizes == [ctypes.sizeof(_lib.ConvDesc), ctypes.sizeof(_lib.BnActDesc), ctypes.sizeof(_lib.RnnParams),
                     ctypes.sizeof(_lib.RnnGrads), ctypes.sizeof(_lib.PackedSeq)]


def test_host_side_errors_without_gpu():
    import pytest
    import torch
    from showtell_amd import ShowTellHipError, ops
    with pytest.raises(ShowTellHipError):
        ops.cast(torch.zeros(4), torch.bfloat16)          # CPU tensor -> loud failure, never a fallback
    from showtell_amd.cnn import ResNet
    with pytest.raises(ValueError):
        ResNet(99)
