"""Build the HIP shared library in-tree:  streamformer_amd/libstreamformer_hip.so

    python streamformer_amd/build.py          # or __graft_entry__.build()   (not `-m`: importing the
                                              #  package needs the library this script produces)

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU box
with the repo snapshot; nothing is JIT-built at import time.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstreamformer_hip.so")
SOURCES = ["sf_gemm.hip", "sf_gemm256.hip", "sf_gemm_panel.hip", "sf_gemm_skinny.hip", "sf_gemm_tile.hip", "sf_switches.hip", "sf_rowwise.hip", "sf_attention.hip", "sf_attention_generic.hip", "sf_pool_head.hip", "sf_loss.hip", "sf_encoder.hip",
           "sf_train_kernels.hip", "sf_wgrad.hip", "sf_attention_bwd.hip", "sf_train.hip"]
# lab library only (build.py --lab): round-4 kernels that were built to parity and did not beat the product path on the wall clock —
# the two epilogue-overlap variants of the panel kernel, the qkv projection with the temporal attention as its epilogue (clip form, round 4;
# streamed-frame form, round 6).
# They live in tools/lab/ (not in the package) and are never part of libstreamformer_hip.so.
LAB_DIR = os.path.join(os.path.dirname(HERE), "tools", "lab")
LAB_SOURCES = ["sf_gemm_pp.hip", "sf_gemm_pipe.hip", "sf_gemm_qkv.hip", "sf_stream_fused.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, lab: bool = False, variant: str = "", variant_flags=()) -> str:
    """variant="x" (tools/ only, A/B builds): the product sources with extra -D flags into libstreamformer_hip_x.so, loaded with SF_LIB=x.
    lab=True: the measurement variant libstreamformer_hip_lab.so (-DSF_LAB: result-discarding timing switches compiled in),
    loaded by tools/ through SF_LIB=lab; never by the package's default path, bench.py or the tests."""
    hipcc = _hipcc()
    objdir = os.path.join(CSRC, "build_lab" if lab else "build")
    lib_path = os.path.join(HERE, "libstreamformer_hip_lab.so") if lab else LIB
    flags = FLAGS + (["-DSF_LAB", "-I" + CSRC] if lab else [])
    if variant:
        objdir = os.path.join(CSRC, "build_" + variant)
        lib_path = os.path.join(HERE, "libstreamformer_hip_%s.so" % variant)
        flags = flags + list(variant_flags)
    sources = SOURCES + (LAB_SOURCES if lab else [])
    os.makedirs(objdir, exist_ok=True)
    if lab and not os.path.isdir(LAB_DIR):
        raise RuntimeError("the lab kernels (tools/lab/) are not present in this checkout")
    headers = [os.path.join(CSRC, "sf_common.h"), os.path.join(CSRC, "sf_train.h"), os.path.join(CSRC, "sf_internal.h"), os.path.join(CSRC, "sf_pool_head.h"), os.path.join(CSRC, "sf_switches.h"),
               os.path.join(os.path.dirname(HERE), "include", "streamformer_hip.h")]

    def compile_one(src: str) -> str:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        srcp = os.path.join(LAB_DIR if src in LAB_SOURCES else CSRC, src)
        if force or _stale(obj, [srcp] + headers):
            cmd = [hipcc, *flags, "-c", srcp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(compile_one, sources))
    if force or _stale(lib_path, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib_path, *objs]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if lab or variant:
        return lib_path
    # the C++ host example of the C ABI (no Python / torch in that process); tests/test_c_host.py runs it on a GPU
    ex_src = os.path.join(os.path.dirname(HERE), "examples", "host_forward.cpp")
    ex_bin = os.path.join(os.path.dirname(HERE), "examples", "host_forward")
    if os.path.exists(ex_src) and (force or _stale(ex_bin, [ex_src, LIB] + headers)):
        cmd = [hipcc, "--offload-arch=gfx950", "-O2", ex_src, "-I" + os.path.join(os.path.dirname(HERE), "include"), "-L" + HERE,
               "-lstreamformer_hip", "-Wl,-rpath,$ORIGIN/../streamformer_amd", "-o", ex_bin]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    _variant = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--variant=")), "")      # --variant=name -DFLAG ...
    print(build(force="--force" in sys.argv, lab="--lab" in sys.argv, variant=_variant, variant_flags=[a for a in sys.argv[1:] if a.startswith("-D")]))
