#!/bin/bash
# round 4, batch P: panel kernel padding rows as zeros (out-of-range DMA offsets) against clamped re-reads of the last row (SF_PANEL_PAD_CLAMP=1)
mkdir -p gpurun_out
L=gpurun_out/r04_panel_pad_zero_ab.txt
: > $L
PP_LAB_ORACLE=1 timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo FAILED >> $L
for i in 1 2 3; do
  SF_PANEL_PAD_CLAMP=1 timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo FAILED >> $L
  SF_X=zero timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo FAILED >> $L
done
grep -v amdgpu.ids $L | sed 's/reproducible=True //'
