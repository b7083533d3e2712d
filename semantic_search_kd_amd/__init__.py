"""Importable alias of the ``semantic-search-kd_amd/`` package directory.

A hyphen cannot appear in a Python module name, so this stub re-points its package
search path at the real directory and runs that package's ``__init__`` in place.
"""
from pathlib import Path as _Path

_REAL = _Path(__file__).resolve().parent.parent / "semantic-search-kd_amd"
__path__ = [str(_REAL)]  # submodules (``.index``, ``._native``, ...) resolve in the real directory
__file__ = str(_REAL / "__init__.py")
exec(compile((_REAL / "__init__.py").read_text(), __file__, "exec"))
