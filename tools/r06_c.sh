#!/bin/bash
# round 6, call C: (1) the 256-column kernel's per-stagger-group tile runs (SF_G256_WALK=15): fabric traffic + forward time;
# (2) the one decisive device-sharing experiment: synthetic victim beside synthetic neighbours (two processes / two streams of
# one process), beside the REAL temporal attention backward, and an RCCL world-size-1 collective as the victim beside it
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
export PYTHONPATH=$R:${PYTHONPATH:-}
cd /tmp && export TMPDIR=/tmp
FWD2="python $R/bench.py --profile --steps 2 --warmup 1"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_w -o w -- $FWD2 > /dev/null 2>&1
for v in "0 0" "15 0" "15 1"; do
  set -- $v
  rm -rf /tmp/pmc_f
  env SF_G256_WALK=$1 $( [ $2 = 1 ] && echo SF_G256_STORE_WT=1 ) rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_f -o f -- $FWD2 > /dev/null 2>&1
  python $R/profiles/pmc_traffic.py /tmp/pmc_f /tmp/pmc_w > $OUT/c_traffic_walk$1_wt$2.json
  echo "== SF_G256_WALK=$1 STORE_WT=$2"
  python - <<PY
import json
d=json.load(open("$OUT/c_traffic_walk$1_wt$2.json"))["kernels"]
for k,v in d.items():
    if "gemm256" in k: print("  ", k[:64], "fetch MB", round(2*v["FETCH_SIZE_KB_avg"]/1024,1), "write MB", round(v["WRITE_SIZE_KB_avg"]/1024,1), "total", round(v["traffic_bytes_corrected"]/1e6,1))
PY
done 2>&1 | tee $OUT/c_traffic_summary.txt
cd $R
for i in 1 2 3; do
  for v in "0 0" "15 0" "15 1"; do
    set -- $v
    echo -n "WALK=$1 WT=$2: "; env SF_G256_WALK=$1 $( [ $2 = 1 ] && echo SF_G256_STORE_WT=1 ) python tools/fwd_time.py 8 16 2>/dev/null
  done
done | tee $OUT/c_walk_fwd_ab.txt
# ---- device sharing ------------------------------------------------------------------------------------------------------------
{
NL=tools/platform/neighbor_lab
echo "### victim alone"; timeout 60 $NL victim 8
for k in 0 1 2 3 4; do
  echo "### two processes, synthetic neighbour mode $k"
  timeout 60 $NL neighbour $k 16 & NP=$!
  sleep 1; timeout 60 $NL victim 12; wait $NP
done
for k in 0 1 2 3 4; do
  echo "### one process, two streams, synthetic neighbour mode $k"; timeout 60 $NL both $k 10
done
echo "### synthetic victim beside the REAL temporal attention backward (tools/noise_ops.py attn_bwd_temporal, another process)"
SF_NOISE_SECONDS=30 SF_NOISE_MODES=attn_bwd_temporal timeout 300 python tools/noise_ops.py > $OUT/c_noise1.log 2>&1 & NP=$!
for i in $(seq 1 200); do grep -q starting $OUT/c_noise1.log 2>/dev/null && break; sleep 1; done; timeout 60 $NL victim 20; wait $NP; tail -2 $OUT/c_noise1.log
echo "### RCCL world-size-1 collectives as the victim beside the REAL temporal attention backward"
SF_NOISE_SECONDS=50 SF_NOISE_MODES=attn_bwd_temporal timeout 300 python tools/noise_ops.py > $OUT/c_noise2.log 2>&1 & NP=$!
for i in $(seq 1 200); do grep -q starting $OUT/c_noise2.log 2>/dev/null && break; sleep 1; done; SF_VICTIM_SECONDS=20 timeout 120 python tools/platform/rccl_victim.py 2>&1 | tail -2; wait $NP; tail -2 $OUT/c_noise2.log
echo "### RCCL world-size-1 collectives alone"
SF_VICTIM_SECONDS=8 timeout 120 python tools/platform/rccl_victim.py 2>&1 | tail -1
} 2>&1 | tee $OUT/c_device_sharing.txt
