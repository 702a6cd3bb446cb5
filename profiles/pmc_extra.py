"""Per-kernel means of a few GRBM / SQ counters from rocprofv3 --pmc passes (one directory per pass):

    python profiles/pmc_extra.py /tmp/px_a /tmp/px_b /tmp/px_c > profiles/rNN_pmc_sq.json

effective_clock_GHz = GRBM_GUI_ACTIVE / dispatch duration of the same (profiled, serialised) dispatch when the csv carries
timestamps; MI355X_MICROARCH.md "DVFS give-back": the chip clocks to its power budget, so this is the clock the MFMA
peak should be scaled by (2.4 GHz nominal).  SQ_* counters are summed over the chip; SQ_WAVE_CYCLES / SQ_ACTIVE_INST_* /
SQ_WAIT_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (guide, price list).
"""
import csv, glob, json, os, sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
args = sys.argv[1:]
note = "rocprofv3 --pmc passes over `bench.py --profile --steps 2 --warmup 1` (B = 8, bf16 mode); means per dispatch"
if args and args[0] == "--note":
    note = args[1]
    args = args[2:]
for d in args:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = row["Kernel_Name"][:100]
                c = row["Counter_Name"]
                a = acc[k][c]
                a[0] += float(row["Counter_Value"]); a[1] += 1
                if c == "GRBM_GUI_ACTIVE" and row.get("Start_Timestamp") and row.get("End_Timestamp"):
                    dur[k][0] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); dur[k][1] += 1
out = {}
for k in sorted(acc):
    e = {c: round(v[0] / max(v[1], 1), 1) for c, v in acc[k].items()}
    e["dispatches"] = max(v[1] for v in acc[k].values())
    if dur[k][1] and "GRBM_GUI_ACTIVE" in e:
        ns = dur[k][0] / dur[k][1]
        e["profiled_duration_us"] = round(ns / 1e3, 2)
        e["effective_clock_GHz_if_counter_is_per_chip"] = round(e["GRBM_GUI_ACTIVE"] / ns, 3)
        e["effective_clock_GHz_if_counter_sums_8_xcds"] = round(e["GRBM_GUI_ACTIVE"] / ns / 8, 3)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "SQ_BUSY_CYCLES" in e and e["SQ_BUSY_CYCLES"]:
        e["mfma_busy_over_sq_busy"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / e["SQ_BUSY_CYCLES"], 4)
    if "SQ_VALU_MFMA_BUSY_CYCLES" in e and e.get("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE sums the 8 XCDs' busy cycles; SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs' cycles with an MFMA in the pipe:
        # fraction of the kernel's SIMD-cycles in which the matrix pipe was busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024)
        e["mfma_busy_frac_of_simd_cycles"] = round(e["SQ_VALU_MFMA_BUSY_CYCLES"] / (e["GRBM_GUI_ACTIVE"] * 128.0), 4)
    if "SQ_ACTIVE_INST_LDS" in e and "SQ_LDS_BANK_CONFLICT" in e and e["SQ_ACTIVE_INST_LDS"]:
        e["lds_bank_conflict_over_active_lds"] = round(e["SQ_LDS_BANK_CONFLICT"] / (4.0 * e["SQ_ACTIVE_INST_LDS"]), 4)   # cycles / (quad-cycles * 4)
    if "SQ_WAIT_INST_ANY" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
        e["wait_inst_any_over_wave_cycles"] = round(e["SQ_WAIT_INST_ANY"] / e["SQ_WAVE_CYCLES"], 4)
    if "SQ_ACTIVE_INST_VALU" in e and "SQ_WAVE_CYCLES" in e and e["SQ_WAVE_CYCLES"]:
        e["valu_active_over_wave_cycles"] = round(e["SQ_ACTIVE_INST_VALU"] / e["SQ_WAVE_CYCLES"], 4)
    out[k] = e
import hashlib
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sha = hashlib.sha256(open(os.path.join(root, "streamformer_amd", "csrc", "sf_gemm_panel.hip"), "rb").read()).hexdigest()[:16]
print(json.dumps({"note": note, "panel_source_sha16": sha, "kernels": out}, indent=1))
