#!/bin/bash
# round 5: training step, side stream on / off with the fused temporal projections (alternating pairs, one box)
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
for i in 1 2 3; do
  for v in 1 0; do
    SF_TRAIN_SIDE_STREAM=$v python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side=$v', d['ms_per_step'], d['value'])"
  done
done | tee $OUT/d_side_stream_ab.txt
