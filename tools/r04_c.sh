#!/bin/bash
# round 4, batch C: persistent double-buffered spatial attention against the two-workgroups-per-CU DMA kernel
mkdir -p gpurun_out
L=gpurun_out/r04_spatial_pers_lab.txt
: > $L
run() { env "$@" timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo "FAILED: $*" >> $L; }
run PP_LAB_ORACLE=1 SF_DISABLE_SPATIAL_PERS=1
run PP_LAB_ORACLE=1 SF_X=pers
run SF_DISABLE_SPATIAL_PERS=1
run SF_X=pers
grep -v amdgpu.ids $L
