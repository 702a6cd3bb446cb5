#!/bin/bash
# training-step kernel table only
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out; TAG=${1:-r03x}
rocprofv3 --kernel-trace -d /tmp/prof_t -o t -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_train_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_t -name "*.db" | head -1) > $OUT/${TAG}_train_step_kernel_stats.txt
head -24 $OUT/${TAG}_train_step_kernel_stats.txt | cut -c1-150
