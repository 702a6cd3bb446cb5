set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{
for lab in 0 8; do SF_TILE_LAB=$lab python tools/tile_lab.py 3136; done
SF_TILE_MAX_M=12544 python tools/tile_lab.py 6272 12544
} > $OUT/r03_k_tile_lab.txt 2>&1
grep -v amdgpu.ids $OUT/r03_k_tile_lab.txt
python -m pytest tests/test_tile_gemm.py tests/test_hip_parity.py -q -m gpu --tb=short -x -k "tile or f2 or base_full or linear" 2>&1 | tail -4
for B in 1 2; do python tools/fwd_time.py $B 16 2>&1 | grep -v amdgpu; done
SF_TILE_MAX_M=6272 python tools/fwd_time.py 2 16 2>&1 | grep -v amdgpu
