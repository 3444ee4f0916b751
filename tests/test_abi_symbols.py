"""The C-ABI library loads and exports every symbol include/keraslm_hip.h declares
(no compute calls: runs without a GPU)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "keraslm_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kl_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from ocrd_keraslm_amd.lib import hipabi
    if not os.path.exists(hipabi.LIB_PATH):
        import __graft_entry__
        __graft_entry__.build()
    lib = C.CDLL(hipabi.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), "library does not export %s" % name
    # the binding table covers the header too
    for name in names:
        assert name in hipabi.SIGNATURES, "hipabi.SIGNATURES lacks %s" % name


def test_host_only_entry_points():
    """configuration / layout calls need no device"""
    from ocrd_keraslm_amd.lib import hipabi
    lib = hipabi.load()
    assert lib.kl_abi_version() == 1
    cfg = hipabi.KlConfig(2, 512, 256, 1, 200, 10)
    assert lib.kl_param_count(C.byref(cfg)) == 4351952
    cfg5 = hipabi.KlConfig(4, 1024, 256, 2, 200, 10)
    assert lib.kl_param_count(C.byref(cfg5)) == 33918880
    bad = hipabi.KlConfig(2, 100, 256, 1, 200, 10)               # width not a multiple of 32
    assert not lib.kl_create(C.byref(bad))
    h = lib.kl_create(C.byref(cfg))
    assert h
    assert lib.kl_derived_bytes(h) > 0
    assert lib.kl_window_workspace_bytes(h, 64, 256, 1) > lib.kl_window_workspace_bytes(h, 64, 256, 0) > 0
    assert lib.kl_prepare(h, 1, None) == 3                       # KL_ERR_STATE: nothing bound
    assert lib.kl_error_string(3).decode().startswith("call order")
    lib.kl_destroy(h)
