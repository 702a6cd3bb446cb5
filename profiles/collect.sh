#!/bin/bash
# Round profiles, run ON the GPU box (gpurun):  bash profiles/collect.sh r02
# kernel-trace and PMC counters are separate rocprofv3 runs (MI355X_MICROARCH.md); outputs land in gpurun_out/<tag>_*.
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_* /tmp/pmc_*
rocprofv3 --kernel-trace -d /tmp/prof_f -o f -- python $R/bench.py --profile --steps 12 --warmup 3 > $OUT/${TAG}_profile_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_f -name "*.db" | head -1) > $OUT/${TAG}_forward_kernel_stats.txt
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- python $R/bench.py --profile --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- python $R/bench.py --profile --steps 2 --warmup 1 > /dev/null 2>&1
python $R/profiles/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w > $OUT/${TAG}_pmc_traffic.json
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > $OUT/${TAG}_streaming_run.txt 2>/dev/null
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S > $OUT/${TAG}_streaming_timeline.txt
python $R/profiles/summarize.py $S > $OUT/${TAG}_streaming_kernel_stats.txt
# PMC traffic of the streaming kernels (two separate passes, two passes over the 64-frame stream each)
SF_REPS=2 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_sf -o f -- python $R/tools/stream_trace.py > /dev/null 2>&1
SF_REPS=2 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_sw -o w -- python $R/tools/stream_trace.py > /dev/null 2>&1
python $R/profiles/pmc_traffic.py /tmp/pmc_sf /tmp/pmc_sw > $OUT/${TAG}_streaming_pmc_traffic.json
# one clip per call (README.md:55-71): kernel table of the B = 1 forward
rocprofv3 --kernel-trace -d /tmp/prof_b1 -o x -- python $R/tools/b1_trace.py 1 > /dev/null 2>&1
python $R/profiles/summarize.py $(find /tmp/prof_b1 -name "*.db" | head -1) > $OUT/${TAG}_b1_forward_kernel_stats.txt
# the fp32-accurate forward
rocprofv3 --kernel-trace -d /tmp/prof_a -o a -- python $R/tools/accurate_fwd.py 10 > $OUT/${TAG}_accurate_run.txt 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_a -name "*.db" | head -1) > $OUT/${TAG}_accurate_forward_kernel_stats.txt
# instruction-shape lab of the 256-column GEMM's phase skeleton
[ -x $R/tools/bin/mfma_shape_lab ] && $R/tools/bin/mfma_shape_lab 20000 > $OUT/${TAG}_mfma_shape_lab.txt 2>&1
rocprofv3 --kernel-trace -d /tmp/prof_t -o t -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_train_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_t -name "*.db" | head -1) > $OUT/${TAG}_train_step_kernel_stats.txt
cd $R && python bench.py > $OUT/${TAG}_bench_line.json 2> $OUT/${TAG}_bench_err.txt
tail -c 600 $OUT/${TAG}_bench_line.json
