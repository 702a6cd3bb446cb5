#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for lab in 0 1 2 3 4 7; do
  rm -rf /tmp/al; SF_ATTN_BWD_LAB=$lab rocprofv3 --kernel-trace --stats -d /tmp/al -o a -- python $R/tools/attn_bwd_lab.py > /dev/null 2>&1
  echo "lab=$lab $(python $R/profiles/summarize.py $(find /tmp/al -name '*.db' | head -1) | grep sf_spatial_attn_bwd | cut -c1-60)"
done
