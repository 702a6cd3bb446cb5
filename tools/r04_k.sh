#!/bin/bash
# round 4, batch K: training step under rocprofv3 (kernel table) at the current HEAD
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_t
timeout 900 rocprofv3 --kernel-trace -d /tmp/prof_t -o t -- python $R/bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline > $R/gpurun_out/r04k_train_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_t -name "*.db" | head -1) > $R/gpurun_out/r04k_train_step_kernel_stats.txt
head -14 $R/gpurun_out/r04k_train_step_kernel_stats.txt | cut -c1-150
tail -c 400 $R/gpurun_out/r04k_train_run_line.json
