#!/bin/bash
# round 5: training step with the temporal branch's two projections fused vs unfused (alternating pairs, one box)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  for v in fused unfused; do
    if [ $v = unfused ]; then export SF_TRAIN_UNFUSED_TEMPORAL=1; else unset SF_TRAIN_UNFUSED_TEMPORAL; fi
    python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['value'], d['losses_per_task_first_last'])"
  done
done | tee $OUT/c_temporal_fuse_ab.txt
