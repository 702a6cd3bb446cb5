#!/bin/bash
# round 4, batch G: spatial attention with the score-tile count as a compile-time constant (NTC = 13) against the run-time count
mkdir -p gpurun_out
L=gpurun_out/r04_spatial_ntc_ab.txt
: > $L
run() { env "$@" timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo "FAILED: $*" >> $L; }
run PP_LAB_ORACLE=1 SF_DISABLE_SPATIAL_NTC=1
run PP_LAB_ORACLE=1 SF_X=ntc
run SF_DISABLE_SPATIAL_NTC=1
run SF_X=ntc
grep -v amdgpu.ids $L
