"""The C-ABI library loads here (no GPU) and exports exactly what include/m355seg.h declares."""
import ctypes
import os
import re
import subprocess

import pytest

from conftest import ROOT
from segmentation_pipeline_amd import _lib


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "m355seg.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(m355_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_what_python_binds():
    assert set(declared_symbols()) == set(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (m355_\w+)", out))
    missing = [s for s in declared_symbols() if s not in exported]
    assert not missing, missing


def test_library_loads_and_reports_version():
    L = _lib.lib()
    assert L.m355_version() == 3 == _lib.ABI_VERSION
    assert L.m355_queue_pool_bytes() == 4096 * 64
    # argument validation of the pool hand-over needs no GPU
    assert L.m355_queue_pool_set(None, 4096 * 64, 0) == -1 and b"queue_pool_set" in L.m355_last_error()
    assert L.m355_queue_pool_set(ctypes.c_void_p(64), 16, 0) == -1
    assert L.m355_last_error() is not None


def test_host_side_queries_and_argument_validation():
    """Pure host logic of the library: workspace queries and argument checks (no launches)."""
    L = _lib.lib()
    d = _lib.ConvDesc(1, 4, 32, 128, 128, 128, 3, 1, 1, 0, 0, 0)
    assert L.m355_conv3d_fwd_workspace(ctypes.byref(d)) >= 4 * 27 * 32 * 4
    assert L.m355_conv3d_bwd_weight_workspace(ctypes.byref(d)) > 0
    nd = _lib.NormDesc(2, 32, 4096, 8, 1, 1e-5, 0.0, 0, 0, 0)
    assert L.m355_norm_num_stats(ctypes.byref(nd)) == 16
    nd_bn = _lib.NormDesc(2, 32, 4096, 0, 1, 1e-5, 0.0, 0, 0, 0)
    assert L.m355_norm_num_stats(ctypes.byref(nd_bn)) == 32
    assert L.m355_norm_workspace(ctypes.byref(nd)) > 0
    assert L.m355_hybrid_loss_workspace(1, 3, 128 ** 3) > 0
    # invalid arguments are rejected before any launch, with a message
    bad = _lib.ConvDesc(0, 4, 32, 8, 8, 8, 3, 1, 1, 0, 0, 0)
    rc = L.m355_conv3d_fwd(ctypes.byref(bad), None, None, None, None, None, None, 0, None)
    assert rc == -1 and b"conv3d_fwd" in L.m355_last_error()
    rc = L.m355_avgpool3d_2x_fwd(ctypes.c_void_p(16), ctypes.c_void_p(16), 1, 1, 3, 4, 4, 0, 0, None)
    assert rc == -2 and b"odd" in L.m355_last_error()
    gn = _lib.NormDesc(1, 30, 64, 8, 0, 1e-5, 0.0, 0, 0, 0)
    rc = L.m355_norm_act_fwd(ctypes.byref(gn), None, None, None, None, None, None, None, None)
    assert rc == -1 and b"divisible" in L.m355_last_error()


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libm355seg.so")
    with pytest.raises(_lib.M355Error, match="no CPU or PyTorch fallback"):
        _lib.lib()
