"""Importable alias of the ``show-tell_amd/`` package directory.

The product directory keeps the project's name (``show-tell_amd``), which is not a
valid Python identifier; ``import showtell_amd`` resolves to it through this shim.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "show-tell_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
