#!/bin/bash
# round 4, batch H: spatial attention BACKWARD with the 32-row block count as a compile-time constant (NTC = 13 -> 7 blocks,
# loops fully unrolled, 0 spills) against the run-time count; parity first, then rocprofv3 kernel stats of tools/attn_bwd_lab.py
mkdir -p gpurun_out
L=gpurun_out/r04_spatial_bwd_ntc_ab.txt
: > $L
[ -n "$SKIP_TESTS" ] || timeout 900 python -m pytest tests/test_train_parity.py -m gpu -x -q -k "attention_bwd or base_model_gradients or dropout_gradients" 2>&1 | tail -5 >> $L
cd /tmp && export TMPDIR=/tmp
for v in off on; do
  if [ $v = off ]; then E="SF_DISABLE_SPATIAL_NTC=1"; else E="SF_X=ntc"; fi
  rm -rf /tmp/prof_$v
  env $E timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$v -o t -- python $GRAFT_REPO_ROOT/tools/attn_bwd_lab.py > /tmp/prof_$v.log 2>&1
  echo "== NTC $v" >> $GRAFT_REPO_ROOT/$L
  python $GRAFT_REPO_ROOT/profiles/summarize.py $(find /tmp/prof_$v -name '*.db' | head -1) | grep sf_spatial_attn_bwd | cut -c1-100 >> $GRAFT_REPO_ROOT/$L
done
cd $GRAFT_REPO_ROOT
grep -v amdgpu.ids $L
