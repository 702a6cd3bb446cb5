#!/bin/bash
# round 4, batch B: the role-split persistent panel kernel (sf_gemm_pipe.hip)
mkdir -p gpurun_out
L=gpurun_out/r04_pipe_lab.txt
: > $L
run() { env SF_LIB=lab "$@" timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo "FAILED: $*" >> $L; }
run PP_LAB_ORACLE=1 SF_X=base
run PP_LAB_ORACLE=1 SF_PANEL_PIPE=1
run SF_PANEL_PIPE=1 SF_PANEL_PIPE_MAX_K=3072
run SF_X=base2
grep -v amdgpu.ids $L
