#!/bin/bash
# round 5: two clips per call — statistics-producing narrow tiles + 256^2 consumers against the in-kernel-statistics tiles
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
python -m pytest tests/test_hip_parity.py -q -m gpu -k "two_clips_run or plane_form or five_clips" > $OUT/i_tests.log 2>&1; tail -4 $OUT/i_tests.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  echo -n "stats tiles + 256^2:  "; python $R/tools/fwd_time.py 2 16 2>/dev/null
  echo -n "in-kernel statistics: "; SF_TILE_FOLD_MIN_M=99999999 python $R/tools/fwd_time.py 2 16 2>/dev/null
done | tee $OUT/i_b2_ab.txt
for T in 20 24 28; do
  echo -n "stats tiles + 256^2:  "; python $R/tools/fwd_time.py 1 $T 2>/dev/null
  echo -n "in-kernel statistics: "; SF_TILE_FOLD_MIN_M=99999999 python $R/tools/fwd_time.py 1 $T 2>/dev/null
done | tee -a $OUT/i_b2_ab.txt
rm -rf /tmp/prof_b
rocprofv3 --kernel-trace -d /tmp/prof_b -o x -- python $R/tools/b1_trace.py 2 > /dev/null 2>&1
python $R/profiles/summarize.py $(find /tmp/prof_b -name "*.db" | head -1) | cut -c1-150 | head -14 | tee $OUT/i_b2_kernel_stats.txt
