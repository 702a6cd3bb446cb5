#!/bin/bash
# round 6, call G: the DEFAULT (fp32-accurate) mode at one / two / four clips per call: kernel tables + wall time
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for b in 1 2 4; do
  rm -rf /tmp/prof_b
  SF_MODE=fp32 rocprofv3 --kernel-trace -d /tmp/prof_b -o x -- python $R/tools/b1_trace.py $b > /dev/null 2>&1
  python $R/profiles/summarize.py $(find /tmp/prof_b -name "*.db" | head -1) | head -16 | cut -c1-170 > $OUT/g_acc_b${b}_kernel_stats.txt
  cat $OUT/g_acc_b${b}_kernel_stats.txt
  cd $R; SF_MODE=fp32 python tools/fwd_time.py $b 16 2>/dev/null; cd /tmp
done
