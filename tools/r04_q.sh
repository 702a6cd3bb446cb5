#!/bin/bash
# round 4, batch Q: accurate (bf16x3) forward with the spatial attention's tile count as a compile-time constant (SF_X=on) against
# the run-time count: alternating wall-clock runs, then the kernel's time under rocprofv3
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
L=$R/gpurun_out/r04_spatial_ntc_acc_ab.txt
: > $L
for i in 1 2 3; do
  echo "off: $(SF_DISABLE_SPATIAL_NTC=1 timeout 300 python tools/accurate_fwd.py 20 2>&1 | grep accurate)" >> $L
  echo "on:  $(timeout 300 python tools/accurate_fwd.py 20 2>&1 | grep accurate)" >> $L
done
cd /tmp && export TMPDIR=/tmp
for v in off on; do
  rm -rf /tmp/pa_$v
  if [ $v = off ]; then E="SF_DISABLE_SPATIAL_NTC=1"; else E="SF_X=on"; fi
  env $E timeout 300 rocprofv3 --kernel-trace -d /tmp/pa_$v -o a -- python $R/tools/accurate_fwd.py 10 > /dev/null 2>&1
  echo "$v: $(python $R/profiles/summarize.py $(find /tmp/pa_$v -name '*.db' | head -1) | grep spatial_attn | cut -c1-120)" >> $L
done
cat $L
