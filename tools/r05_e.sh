#!/bin/bash
# round 5: kernel table of the training step (side stream off, so that per-kernel durations are not inflated by the overlap)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
SF_TRAIN_SIDE_STREAM=0 rocprofv3 --kernel-trace -d /tmp/prof_t -o t -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $OUT/e_train_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_t -name "*.db" | head -1) > $OUT/e_train_step_kernel_stats_noside.txt
head -45 $OUT/e_train_step_kernel_stats_noside.txt | cut -c1-150
