"""CPU: the C-ABI library loads and exports every symbol include/streamformer_hip.h declares (no
compute calls), and the ctypes table covers exactly that set."""
import ctypes
import os
import re

from tests.conftest import ROOT


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "streamformer_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(sf_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    import streamformer_amd._native as nat
    lib = ctypes.CDLL(nat.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in the header but not exported"
    assert sorted(nat.SIGNATURES) == syms, "ctypes table and header disagree"
    assert nat.lib.sf_abi_version() == 4


def test_create_validates_config_without_gpu():
    import streamformer_amd._native as nat
    cfg = nat.SfConfig(224, 16, 3, 16, 768, 12, 12, 3072, 0, 1, 1, 0, 1e-6)
    h = ctypes.c_void_p()
    assert nat.lib.sf_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == 0
    assert nat.lib.sf_missing_weights(h) > 200          # nothing loaded yet; message lists names
    assert b"missing" in nat.lib.sf_last_error()
    nat.lib.sf_destroy(h)
    bad = nat.SfConfig(224, 16, 3, 16, 768, 12, 8, 3072, 0, 1, 1, 0, 1e-6)   # head_dim 96
    assert nat.lib.sf_create(ctypes.byref(bad), 0, ctypes.byref(h)) == nat.SF_ERR_INVALID
    assert b"head_dim" in nat.lib.sf_last_error()


def test_product_library_holds_no_result_discarding_lab_switches():
    """timing switches that skip stores / phases are compiled in only with -DSF_LAB (build.py --lab); the product
    .so must not even contain their names."""
    import streamformer_amd._native as nat
    blob = open(os.path.join(os.path.dirname(nat.LIB_PATH), "libstreamformer_hip.so"), "rb").read()
    names = set(re.findall(rb"SF_[A-Z0-9_]*LAB[A-Z0-9_]*", blob))
    assert not names, f"lab switches compiled into the product library: {sorted(names)}"
