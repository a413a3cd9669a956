"""The C-ABI library loads without a GPU and exports every symbol include/pips_hip.h declares."""
import ctypes
import os
import re

import pips_ipmpp_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "pips_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pips_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported():
    syms = declared_symbols()
    assert len(syms) >= 50
    lib = ctypes.CDLL(pa.capi.LIB_PATH)
    missing = [s for s in syms if not hasattr(lib, s)]
    assert not missing, missing


def test_binding_list_matches_header():
    assert sorted(pa.capi.SYMBOLS) == declared_symbols()


def test_no_gpu_calls_fail_loudly_not_silently():
    """Without a device every compute entry point returns an error code; there is no CPU fallback."""
    import numpy as np
    if pa.device_count() > 0:
        return
    K, _ = pa.kkt_leaf_assemble(4, pa.Csr(2, 4, [0, 2, 4], [0, 1, 2, 3], [1.0, 2.0, 3.0, 4.0]))
    s = pa.HipLdlSolver(K, n_primal=4)
    try:
        s.matrixChanged()
        raised = False
    except pa.capi.PipsHipError as e:
        raised = "no HIP device" in str(e) or "failed" in str(e)
    assert raised


def test_library_does_not_link_the_oracle():
    out = os.popen(f"readelf -d {pa.capi.LIB_PATH}").read()
    assert "oracle" not in out
