#!/bin/bash
# round 4, batch J: LayerNorm backward — grid size (workgroups of 4 waves; 512 = round 3) with the accumulated-gradient row loaded up front.
# Kernel time from rocprofv3 (the op entry allocates its partial-sum buffer per call, so wall time says nothing).
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
L=$R/gpurun_out/r04_ln_bwd_lab.txt
: > $L
cd /tmp && export TMPDIR=/tmp
for b in 512 1024 1536 2048; do
  rm -rf /tmp/lnb
  SF_LN_BWD_BLOCKS=$b timeout 120 rocprofv3 --kernel-trace -d /tmp/lnb -o a -- python $R/tools/ln_bwd_lab.py > /dev/null 2>&1
  echo "blocks=$b $(python $R/profiles/summarize.py $(find /tmp/lnb -name '*.db' | head -1) | grep 'sf_ln_bwd' | cut -c1-110)" >> $L
done
cd $R
[ -n "$SKIP_TESTS" ] || timeout 600 python -m pytest tests/test_train_parity.py -m gpu -x -q -k "layernorm_bwd or small_model_gradients" 2>&1 | tail -3 >> $L
grep -v amdgpu.ids $L
