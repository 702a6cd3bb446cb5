#!/bin/bash
# round 4, batch I: is the K = 768 panel kernel's epilogue bound by the chip (HBM) or per CU?  Lab library, SF_PANEL_LAB_EPI:
# 0 = product, 1 = half of the tiles skip their epilogue, 3 = three of four skip, 2 = all skip (main loop alone).  Results invalid by design.
mkdir -p gpurun_out
L=gpurun_out/r04_panel_epi_share_lab.txt
: > $L
for m in 0 1 3 2; do
  SF_LIB=lab SF_PANEL_LAB_EPI=$m timeout 300 python tools/pp_lab.py >> $L 2>&1 || echo "FAILED $m" >> $L
done
grep -v amdgpu.ids $L | sed 's/checksum.*|/|/'
