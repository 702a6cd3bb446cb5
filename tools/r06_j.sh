#!/bin/bash
# round 6, call J: the attention kernels' 128-byte-row image swizzle (bswz / sp_bswz without row bit 3) against the old function
# (build.py --variant=aswzold -DSF_ATTN_SWZ_LEGACY): parity, kernel times, forward, training step, conflict counters
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -q -m gpu -x -k "attention or attn or golden or base_model_gradients_match_oracle_at or small_model_gradients" > $OUT/j_tests.log 2>&1; tail -3 $OUT/j_tests.log
for i in 1 2 3; do
  echo -n "new: "; python tools/attn_time.py 2>/dev/null
  echo -n "old: "; SF_LIB=aswzold python tools/attn_time.py 2>/dev/null
done | tee $OUT/j_attn_ab.txt
for i in 1 2; do
  echo -n "new fwd "; python tools/fwd_time.py 8 16 2>/dev/null
  echo -n "old fwd "; SF_LIB=aswzold python tools/fwd_time.py 8 16 2>/dev/null
  echo -n "new acc "; SF_MODE=fp32 python tools/fwd_time.py 8 16 2>/dev/null
  echo -n "old acc "; SF_MODE=fp32 SF_LIB=aswzold python tools/fwd_time.py 8 16 2>/dev/null
done | tee -a $OUT/j_attn_ab.txt
for i in 1 2; do
  echo -n "new train "; python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
  echo -n "old train "; SF_LIB=aswzold python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.read().strip().splitlines()[-1])['ms_per_step'])"
done | tee -a $OUT/j_attn_ab.txt
