#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -d /tmp/wl -o w -- python $R/tools/wgrad_lab.py > $R/gpurun_out/wgrad_lab.log 2>&1
python $R/profiles/summarize.py $(find /tmp/wl -name "*.db" | head -1) > $R/gpurun_out/wgrad_lab_kernels.txt 2>&1
