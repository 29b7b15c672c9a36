"""show-tell on MI355X: HIP/CDNA4 kernels behind the reference's nn.Module surface."""
from . import _lib  # noqa: F401
from ._lib import ShowTellHipError  # noqa: F401

__all__ = ["ShowTellHipError"]
