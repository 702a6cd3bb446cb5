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
    assert nat.lib.sf_abi_version() == 5


def test_create_validates_config_without_gpu():
    import streamformer_amd._native as nat
    cfg = nat.SfConfig(224, 16, 3, 16, 768, 12, 12, 3072, 0, 1, 1, 0, 1e-6)
    h = ctypes.c_void_p()
    assert nat.lib.sf_create(ctypes.byref(cfg), 0, ctypes.byref(h)) == 0
    assert nat.lib.sf_missing_weights(h) > 200          # nothing loaded yet; message lists names
    assert b"missing" in nat.lib.sf_last_error()
    nat.lib.sf_destroy(h)
    # round 6: head widths other than 64 are accepted (multiples of 8 up to 128), any intermediate / patch size; what is still refused says why
    for ok in (nat.SfConfig(224, 16, 3, 16, 768, 12, 8, 3072, 0, 1, 1, 0, 1e-6),          # head_dim 96
               nat.SfConfig(224, 14, 3, 16, 1152, 27, 16, 4304, 0, 1, 1, 0, 1e-6)):      # SigLIP-so400m: head_dim 72, I = 4304, 14 x 14 patches
        assert nat.lib.sf_create(ctypes.byref(ok), 0, ctypes.byref(h)) == 0
        nat.lib.sf_destroy(h)
    for bad in (nat.SfConfig(224, 16, 3, 16, 320, 2, 2, 640, 0, 1, 1, 0, 1e-6),          # head_dim 160 > 128
                nat.SfConfig(224, 16, 3, 16, 192, 2, 16, 384, 0, 1, 1, 0, 1e-6)):        # head_dim 12: not a multiple of 8
        assert nat.lib.sf_create(ctypes.byref(bad), 0, ctypes.byref(h)) == nat.SF_ERR_INVALID
        assert b"head_dim" in nat.lib.sf_last_error()
    narrow = nat.SfConfig(224, 16, 3, 16, 144, 2, 2, 304, 0, 1, 1, 0, 1e-6)               # hidden_size not a multiple of 64
    assert nat.lib.sf_create(ctypes.byref(narrow), 0, ctypes.byref(h)) == nat.SF_ERR_INVALID and b"hidden_size" in nat.lib.sf_last_error()
    wide = nat.SfConfig(224, 16, 3, 16, 1280, 2, 20, 5120, 0, 1, 1, 0, 1e-6)       # 20 heads of 64
    assert nat.lib.sf_create(ctypes.byref(wide), 0, ctypes.byref(h)) == nat.SF_ERR_INVALID and b"heads" in nat.lib.sf_last_error()


def test_product_library_holds_no_result_discarding_lab_switches():
    """timing switches that skip stores / phases are compiled in only with -DSF_LAB (build.py --lab); the product
    .so must not even contain their names."""
    import streamformer_amd._native as nat
    blob = open(os.path.join(os.path.dirname(nat.LIB_PATH), "libstreamformer_hip.so"), "rb").read()
    names = set(re.findall(rb"SF_[A-Z0-9_]*LAB[A-Z0-9_]*", blob))
    assert not names, f"lab switches compiled into the product library: {sorted(names)}"


def test_product_library_exports_exactly_the_header():
    """the dynamic symbol table of the product .so holds the header's `sf_*` entry points and nothing else named
    `sf_*` with C linkage — in particular none of the lab kernels (`*_pp*`, `*_pipe*`, `*_lab*`; tools/lab/, build.py --lab)."""
    import subprocess
    import streamformer_amd._native as nat
    nm = "/opt/rocm/lib/llvm/bin/llvm-nm"
    if not os.path.exists(nm):
        nm = "nm"
    out = subprocess.run([nm, "-D", "--defined-only", nat.LIB_PATH], capture_output=True, text=True, check=True).stdout
    syms = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    c_abi = sorted(s for s in syms if re.fullmatch(r"sf_[a-z0-9_]+", s))
    assert c_abi == declared_symbols(), sorted(set(c_abi) ^ set(declared_symbols()))
    bad = [s for s in syms if re.search(r"_pp_|gemm_pp|_pipe|_lab|gemm_qkv|spatial_attn_pers", s)]
    assert not bad, bad[:5]


def test_switch_table_is_the_only_reader_of_the_environment():
    """One table (csrc/sf_switches.h), one getenv call site (csrc/sf_switches.hip): every switch has a name and a description,
    sf_reload_switches() exists for in-process flips, and no other product source calls getenv."""
    import glob
    import streamformer_amd._native as nat
    names = []
    i = 0
    while True:
        n = nat.lib.sf_switch_info(i, 0)
        if n is None:
            break
        d = nat.lib.sf_switch_info(i, 1)
        assert n.startswith(b"SF_") and d and len(d) > 8, (n, d)
        names.append(n.decode())
        i += 1
    assert len(names) >= 30 and len(set(names)) == len(names)
    nat.lib.sf_reload_switches()
    csrc = os.path.join(ROOT, "streamformer_amd", "csrc")
    total = 0
    for f in glob.glob(os.path.join(csrc, "*.hip")):
        n = len(re.findall(r"\bgetenv\(", open(f).read()))
        assert n == 0 or f.endswith("sf_switches.hip"), f
        total += n
    assert total <= 5
