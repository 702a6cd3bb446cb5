#!/bin/bash
# round 5: a ViT-L/16-shaped encoder on the generic kernels (frames/s, accuracy, kernel table)
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/vitl_forward.py 8 10 2>/dev/null | tee $OUT/g_vitl_forward.txt
rm -rf /tmp/prof_v
SF_VITL_ORACLE=0 rocprofv3 --kernel-trace -d /tmp/prof_v -o v -- python $R/tools/vitl_forward.py 8 4 > /dev/null 2>&1
python $R/profiles/summarize.py $(find /tmp/prof_v -name "*.db" | head -1) | head -24 | cut -c1-150 >> $OUT/g_vitl_forward.txt
tail -26 $OUT/g_vitl_forward.txt
