#!/bin/bash
# round 6, call F: the so400m-shaped encoder on the generic-width path: frames/s, error, kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
timeout 900 python tools/so400m_forward.py 4 5 2>&1 | grep -v amdgpu.ids | tee $OUT/f_so400m.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_x
SF_SO400M_ORACLE=0 rocprofv3 --kernel-trace -d /tmp/prof_x -o x -- python $R/tools/so400m_forward.py 4 3 > /dev/null 2>&1
python $R/profiles/summarize.py $(find /tmp/prof_x -name "*.db" | head -1) | cut -c1-200 | head -30 | tee $OUT/f_so400m_kernel_stats.txt
