#!/bin/bash
# round 6, call B: (1) SQ counters of the forward at HEAD (LDS bank conflicts after the sf_swz64 fix), (2) fabric traffic + forward time of the
# 256-column kernel's tile walks (SF_G256_WALK = column-group width, SF_G256_STORE_WT = sc1 output stores)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
export PYTHONPATH=$R:${PYTHONPATH:-}
cd /tmp && export TMPDIR=/tmp
FWD2="python $R/bench.py --profile --steps 2 --warmup 1"
rm -rf /tmp/px_a /tmp/px_b /tmp/px_c
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/px_a -o a -- $FWD2 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/px_b -o b -- $FWD2 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d /tmp/px_c -o c -- $FWD2 > /dev/null 2>&1
python $R/profiles/pmc_extra.py --note "round 6 HEAD after the sf_swz64 fix: rocprofv3 --pmc passes (GRBM_GUI_ACTIVE | SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU | SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY) over bench.py --profile --steps 2 --warmup 1: 8-clip bf16 forward; means per dispatch" /tmp/px_a /tmp/px_b /tmp/px_c > $OUT/b_pmc_sq.json
python - <<PY
import json
d=json.load(open("$OUT/b_pmc_sq.json"))["kernels"]
for k,v in d.items():
    if "gemm" in k or "attn" in k: print(k[:70], "conflict/active", v.get("lds_bank_conflict_over_active_lds"), "mfma_busy", v.get("mfma_busy_frac_of_simd_cycles"), "us", v.get("profiled_duration_us"))
PY
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- $FWD2 > /dev/null 2>&1
for v in "0 0" "5 0" "5 1" "0 1" "3 0" "3 1" "4 1" "6 1"; do
  set -- $v
  rm -rf /tmp/pmc_f
  env SF_G256_WALK=$1 $( [ $2 = 1 ] && echo SF_G256_STORE_WT=1 ) rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- $FWD2 > /dev/null 2>&1
  python $R/profiles/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w > $OUT/b_traffic_walk$1_wt$2.json
  echo "== SF_G256_WALK=$1 STORE_WT=$2"
  python - <<PY
import json
d=json.load(open("$OUT/b_traffic_walk$1_wt$2.json"))["kernels"]
for k,v in d.items():
    if "gemm256" in k or "panel" in k: print("  ", k[:64], "fetch MB", round(2*v["FETCH_SIZE_KB_avg"]/1024,1), "write MB", round(v["WRITE_SIZE_KB_avg"]/1024,1), "total", round(v["traffic_bytes_corrected"]/1e6,1))
PY
done 2>&1 | tee $OUT/b_traffic_summary.txt
cd $R
for i in 1 2 3; do
  for v in "0 0" "5 0" "5 1" "0 1" "3 1" "4 1" "6 1"; do
    set -- $v
    echo -n "WALK=$1 WT=$2: "; env SF_G256_WALK=$1 $( [ $2 = 1 ] && echo SF_G256_STORE_WT=1 ) python tools/fwd_time.py 8 16 2>/dev/null
  done
done | tee $OUT/b_walk_fwd_ab.txt
