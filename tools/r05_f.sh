#!/bin/bash
# round 5: streaming p50 under a few existing switches (tile family of the big-N GEMMs of a streamed frame)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
mkdir -p gpurun_out/r05
for i in 1 2; do
  echo "default:            $(SF_REPS=4 python tools/stream_trace.py)"
  echo "GEMM_MID_MIN_M=128: $(SF_GEMM_MID_MIN_M=128 SF_REPS=4 python tools/stream_trace.py)"
  echo "SKINNY_TPS=2:       $(SF_SKINNY_TPS=2 SF_REPS=4 python tools/stream_trace.py)"
  echo "SKINNY_TPS=4:       $(SF_SKINNY_TPS=4 SF_REPS=4 python tools/stream_trace.py)"
done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05/f_stream_switches.txt
