"""HBM-side bytes per launch from two rocprofv3 PMC passes (separate runs, as MI355X_MICROARCH.md prescribes):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python bench.py --profile --steps 2 --warmup 1
    rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python bench.py --profile --steps 2 --warmup 1
    python profiles/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w > profiles/rNN_pmc_traffic.json

Units: FETCH_SIZE / WRITE_SIZE count KB per dispatch.  gfx950 correction (guide, HBM section): FETCH_SIZE
under-counts wide coalesced reads by exactly 2x -> read bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE as is.
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def load(d, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row.get("Counter_Name") != counter:
                    continue
                k = row["Kernel_Name"]
                acc[k][0] += float(row["Counter_Value"])
                acc[k][1] += 1
    return acc


def main(df, dw):
    f, w = load(df, "FETCH_SIZE"), load(dw, "WRITE_SIZE")
    out = {}
    for k in sorted(set(f) | set(w)):
        fk = f[k][0] / max(f[k][1], 1)
        wk = w[k][0] / max(w[k][1], 1)
        out[k[:110]] = {"FETCH_SIZE_KB_avg": round(fk, 1), "dispatches": f[k][1] or w[k][1], "WRITE_SIZE_KB_avg": round(wk, 1),
                        "traffic_bytes_corrected": int(2 * fk * 1024 + wk * 1024)}
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sha = hashlib.sha256(open(os.path.join(root, "streamformer_amd", "csrc", "sf_gemm_panel.hip"), "rb").read()).hexdigest()[:16]
    print(json.dumps({"panel_source_sha16": sha, "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes, bench.py --profile --steps 2 "
                              "--warmup 1, B=8 bf16 mode. Units: KB per dispatch (mean). Correction per MI355X_MICROARCH.md HBM "
                              "section: read_bytes = 2 * FETCH_SIZE * 1024 on gfx950; WRITE_SIZE as is. FETCH counts L2 misses at "
                              "the fabric, Infinity-Cache hits included.", "kernels": out}, indent=1))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
