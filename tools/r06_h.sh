#!/bin/bash
# round 6, call H: fp32-accurate mode at 1 / 2 / 4 clips: folded plane form on the 256^2 kernel (default) against fp32 residual + standalone LayerNorm
# with the N = 768 projections on the 128^2 kernel (more tiles at small M)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
for b in 1 2 4; do
  for i in 1 2; do
    echo -n "default                                   "; SF_MODE=fp32 python tools/fwd_time.py $b 16 2>/dev/null
    echo -n "SF_DISABLE_ACC_FOLD=1                     "; SF_DISABLE_ACC_FOLD=1 SF_MODE=fp32 python tools/fwd_time.py $b 16 2>/dev/null
    echo -n "SF_DISABLE_ACC_FOLD=1 G256_SPLIT_MIN_N=1024 "; SF_DISABLE_ACC_FOLD=1 SF_G256_SPLIT_MIN_N=1024 SF_MODE=fp32 python tools/fwd_time.py $b 16 2>/dev/null
    echo -n "SF_DISABLE_G256_SPLIT=1 (all on 128^2)    "; SF_DISABLE_ACC_FOLD=1 SF_DISABLE_G256_SPLIT=1 SF_MODE=fp32 python tools/fwd_time.py $b 16 2>/dev/null
  done
done | tee gpurun_out/r06/h_acc_small_batch_ab.txt
