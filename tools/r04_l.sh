#!/bin/bash
# round 4, batch L: kernel tables of the forward at 2 and 4 clips per call
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for b in 2 4; do
  rm -rf /tmp/prof_b$b
  timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_b$b -o x -- python $R/tools/b1_trace.py $b > /dev/null 2>&1
  python $R/profiles/summarize.py $(find /tmp/prof_b$b -name "*.db" | head -1) > $R/gpurun_out/r04_b${b}_forward_kernel_stats.txt
  head -14 $R/gpurun_out/r04_b${b}_forward_kernel_stats.txt | cut -c1-150
done
