#!/bin/bash
# round 6, call A: the 64-byte-row LDS swizzle (sf_swz64): device microbench over all 256 functions, parity of the Linear layers and the
# BASELINE-batch forward, then alternating forward / per-shape GEMM timings against the rounds-1-5 function (SF_LIB=swzold), and the vendor
# library on the same shapes
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
timeout 120 tools/bin/lds_swizzle_lab | tee $OUT/a_lds_swizzle_lab.txt
timeout 900 python -m pytest tests -q -m gpu -x -k "op_linear or baseline_batch8 or golden or full_tensor" > $OUT/a_tests.log 2>&1; tail -4 $OUT/a_tests.log
for i in 1 2 3; do
  echo -n "new  bf16 "; python tools/fwd_time.py 8 16
  echo -n "old  bf16 "; SF_LIB=swzold python tools/fwd_time.py 8 16
done 2>&1 | tee $OUT/a_fwd_ab.txt
for i in 1 2; do
  echo -n "new  fp32 "; SF_MODE=fp32 python tools/fwd_time.py 8 16
  echo -n "old  fp32 "; SF_MODE=fp32 SF_LIB=swzold python tools/fwd_time.py 8 16
done 2>&1 | tee -a $OUT/a_fwd_ab.txt
for i in 1 2; do
  echo -n "new: "; python tools/gemm_shapes.py
  echo -n "old: "; SF_LIB=swzold python tools/gemm_shapes.py
done 2>&1 | tee $OUT/a_gemm_shapes.txt
python tools/blas_compare.py 2>&1 | tee $OUT/a_blas.txt
for b in 1 4; do
  echo -n "new B=$b "; python tools/fwd_time.py $b 16
  echo -n "old B=$b "; SF_LIB=swzold python tools/fwd_time.py $b 16
done 2>&1 | tee $OUT/a_small_b.txt
