"""Importable alias of the `pips-ipmpp_amd/` package directory (a hyphen cannot be imported).

The product is the C-ABI shared library `pips-ipmpp_amd/libpipship.so` (HIP kernels + C++ host, sources under
`pips-ipmpp_amd/csrc/`); the Python modules under `pips-ipmpp_amd/python/` are a ctypes mirror of the reference's
DoubleLinearSolver plug-in surface used by the tests and bench.py.
"""
import os as _os

_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
__path__.append(_os.path.join(_root, "pips-ipmpp_amd", "python"))

from .capi import *  # noqa: F401,F403,E402
from . import capi  # noqa: E402,F401
