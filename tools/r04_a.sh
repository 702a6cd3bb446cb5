#!/bin/bash
# round 4, batch A: the ping-pong panel kernel against the one-tile-per-CU panel kernel
mkdir -p gpurun_out
L=gpurun_out/r04_pp_lab.txt
: > $L
run() { env SF_LIB=lab "$@" timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo "FAILED: $*" >> $L; }
run PP_LAB_ORACLE=1 SF_X=base
run PP_LAB_ORACLE=1 SF_PANEL_PP=1
run SF_PANEL_PP=1 SF_PANEL_PP_STAGGER_NS=0
run SF_PANEL_PP=1 SF_PANEL_PP_STAGGER_MODE=2
run SF_PANEL_PP=1 SF_PANEL_PP_STAGGER_NS=4000
run SF_PANEL_PP=1 SF_PANEL_PP_STAGGER_NS=12000
run SF_PANEL_PP=1 SF_PANEL_PP_MAX_K=3072
run SF_X=base2
cat $L
