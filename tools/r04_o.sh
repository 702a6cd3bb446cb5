#!/bin/bash
# round 4, batch O (lab library): alternating wall-clock A/B of the 8-clip forward — product schedule / qkv projection + temporal attention in one
# launch (SF_QKV_FUSED=1) / that plus the spatial qkv projection on the panel tile (SF_SQKV_PANEL=1); then the kernel tables (rocprofv3) of the first two
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
L=$R/gpurun_out/r04_qkv_fused_ab.txt
: > $L
for i in 1 2 3; do
  SF_LIB=lab timeout 300 python tools/pp_lab.py >> $L 2>&1
  SF_LIB=lab SF_QKV_FUSED=1 timeout 300 python tools/pp_lab.py >> $L 2>&1
  SF_LIB=lab SF_QKV_FUSED=1 SF_SQKV_PANEL=1 timeout 300 python tools/pp_lab.py >> $L 2>&1
done
cd /tmp && export TMPDIR=/tmp
for v in off on; do
  rm -rf /tmp/pq_$v
  if [ $v = off ]; then E="SF_X=off"; else E="SF_QKV_FUSED=1"; fi
  env SF_LIB=lab $E timeout 300 rocprofv3 --kernel-trace -d /tmp/pq_$v -o x -- python $R/tools/b1_trace.py 8 > /dev/null 2>&1
  echo "== kernel table, fused $v" >> $L
  python $R/profiles/summarize.py $(find /tmp/pq_$v -name "*.db" | head -1) | head -9 | cut -c1-140 >> $L
done
grep -v amdgpu.ids $L | sed 's/reproducible.*|/|/'
