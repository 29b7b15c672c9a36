"""Load the reference's decoder classes in the authoring container (TEST INFRASTRUCTURE).

Used only by ``oracle/gen_golden.py`` and ``tests/test_oracle_pin.py``; both
skip when ``/root/reference`` is absent (it never travels to the GPU box).

The reference files are imported unmodified.  ``rnn.py`` imports torchvision
(absent here) at module top although ``RNN`` itself only needs ``torch.nn``, so
empty stub modules are registered first.  ``rnn_attn.py``
hard-codes ``.cuda()`` (rnn_attn.py:64,65,128); ``cpu_cuda()`` patches
``Tensor.cuda`` to the identity for the duration of a call.
"""
import contextlib
import importlib.util
import os
import sys
import types

REF = os.environ.get("SHOWTELL_REFERENCE", "/root/reference")


def available():
    return os.path.isfile(os.path.join(REF, "rnn.py"))


def _stub_torchvision():
    for name in ("torchvision", "torchvision.models", "torchvision.transforms", "torchvision.datasets"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    tv = sys.modules["torchvision"]
    tv.models = sys.modules["torchvision.models"]
    tv.transforms = sys.modules["torchvision.transforms"]
    tv.datasets = sys.modules["torchvision.datasets"]


def _load(relpath, modname):
    sys.dont_write_bytecode = True
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def load_reference():
    """Returns a namespace with the reference classes/functions on the hot path."""
    _stub_torchvision()
    if REF not in sys.path:
        sys.path.insert(0, REF)  # rnn.py does `from cnn import ResNet`
    ns = types.SimpleNamespace()
    ns.rnn = _load("rnn.py", "_ref_rnn")
    ns.rnn_lstm = _load("LSTM/rnn_lstm.py", "_ref_rnn_lstm")
    ns.rnn_attn = _load("Attention/rnn_attn.py", "_ref_rnn_attn")
    ns.rnn_attn_lstm = _load("Attention/rnn_attn_LSTM.py", "_ref_rnn_attn_lstm")
    ns.beam_search = _load("beam_search.py", "_ref_beam_search")
    ns.metrics = _load("evaluation/evaluation_metrics.py", "_ref_eval_metrics")
    return ns


@contextlib.contextmanager
def cpu_cuda():
    import torch
    orig = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        yield
    finally:
        torch.Tensor.cuda = orig
