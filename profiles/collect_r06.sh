#!/bin/bash
# Round-6 profile set, run ON the GPU box (gpurun):  bash profiles/collect_r06.sh [tag]
# kernel traces and PMC counters are separate rocprofv3 runs, one counter group per run (MI355X_MICROARCH.md); outputs in gpurun_out/<tag>_*.
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
export PYTHONPATH=$R:${PYTHONPATH:-}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_* /tmp/pmc_* /tmp/px_*
FWD="python $R/bench.py --profile --steps 12 --warmup 3"
FWD2="python $R/bench.py --profile --steps 2 --warmup 1"
TRN="python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline"
TRN2="python $R/bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline"
ACC="python $R/tools/accurate_fwd.py 10"
ACC2="python $R/tools/accurate_fwd.py 2"
# ---- kernel tables -------------------------------------------------------------------------------------------------
rocprofv3 --kernel-trace -d /tmp/prof_f -o f -- $FWD > $OUT/${TAG}_profile_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_f -name "*.db" | head -1) > $OUT/${TAG}_forward_kernel_stats.txt
rocprofv3 --kernel-trace -d /tmp/prof_t -o t -- $TRN > $OUT/${TAG}_train_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_t -name "*.db" | head -1) > $OUT/${TAG}_train_step_kernel_stats.txt
rocprofv3 --kernel-trace -d /tmp/prof_a -o a -- $ACC > $OUT/${TAG}_accurate_run.txt 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_a -name "*.db" | head -1) > $OUT/${TAG}_accurate_forward_kernel_stats.txt
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > $OUT/${TAG}_streaming_run.txt 2>/dev/null
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S > $OUT/${TAG}_streaming_timeline.txt
python $R/profiles/summarize.py $S > $OUT/${TAG}_streaming_kernel_stats.txt
for b in 1 2 4; do
  rm -rf /tmp/prof_b
  rocprofv3 --kernel-trace -d /tmp/prof_b -o x -- python $R/tools/b1_trace.py $b > /dev/null 2>&1
  python $R/profiles/summarize.py $(find /tmp/prof_b -name "*.db" | head -1) > $OUT/${TAG}_b${b}_forward_kernel_stats.txt
done
# ---- HBM-side traffic (FETCH_SIZE / WRITE_SIZE in separate passes) -------------------------------------------------
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- $FWD2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- $FWD2 > /dev/null 2>&1
python $R/profiles/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w > $OUT/${TAG}_pmc_traffic.json
# ---- SQ / GRBM counters at HEAD: forward, training step, accurate mode (three passes each) ---------------------------
sq() {   # $1 = name, $2 = note, rest = command
  local name=$1 note=$2; shift 2
  rm -rf /tmp/px_a /tmp/px_b /tmp/px_c
  rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/px_a -o a -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/px_b -o b -- "$@" > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d /tmp/px_c -o c -- "$@" > /dev/null 2>&1
  python $R/profiles/pmc_extra.py --note "$note" /tmp/px_a /tmp/px_b /tmp/px_c > $OUT/${TAG}_pmc_sq${name}.json
}
sq "" "rocprofv3 --pmc passes (GRBM_GUI_ACTIVE | SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU | SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY) over bench.py --profile --steps 2 --warmup 1: 8-clip bf16 forward; means per dispatch" $FWD2
sq "_train" "same three passes over bench.py --mode train --steps 2 --warmup 1: the 8-clip multitask training step; means per dispatch" $TRN2
sq "_accurate" "same three passes over tools/accurate_fwd.py 2: the 8-clip fp32-accurate (bf16x3) forward; means per dispatch" $ACC2
cd $R && python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_err.txt
tail -c 400 $OUT/${TAG}_bench_line.json
