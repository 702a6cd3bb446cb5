"""AccVGPR allocation of every kernel of the library (compile each unit to assembly with the flags build.py uses, read the
code-object metadata).  Kernels listed here keep live values in AccVGPRs; DESIGN.md 4 ("Repeatability under device sharing")
explains why that matters when several processes share one device.   python tools/agpr_report.py [unit.hip ...]"""
import os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "streamformer_amd"))
import build as B          # noqa: E402  (the build script, not the package)

def report(src):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "u.s")
        flags = [f for f in B.FLAGS if f not in ("-fPIC",)] + getattr(B, "EXTRA_FLAGS", {}).get(src, [])
        subprocess.run([B._hipcc(), *flags, "--cuda-device-only", "-S", os.path.join(B.CSRC, src), "-o", out], check=True, stderr=subprocess.DEVNULL)
        txt = open(out).read()
    rows = []
    for m in re.finditer(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.vgpr_count:\s+(\d+)", txt, re.S):
        rows.append((int(m.group(1)), int(m.group(3)), m.group(2)))
    return src, rows

units = sys.argv[1:] or [s for s in B.SOURCES if s not in ("sf_switches.hip", "sf_encoder.hip", "sf_train.hip")]
with ThreadPoolExecutor(max_workers=8) as ex:
    for src, rows in ex.map(report, units):
        used = [r for r in rows if r[0] > 0]
        print(f"{src}: {len(rows)} kernels, {len(used)} with AccVGPRs")
        for a, v, n in used:
            try:
                demangled = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
            except OSError:
                demangled = n
            print(f"    acc {a:3d} of {v:3d} registers  {demangled[:110]}")
