set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{
SF_DISABLE_GEMM_TILE=1 python tools/tile_lab.py 25088 12544
SF_TILE_MAX_M=30000 SF_TILE_SHAPE=6 python tools/tile_lab.py 25088 12544 6272
SF_TILE_MAX_M=30000 SF_TILE_SHAPE=1 python tools/tile_lab.py 25088
SF_TILE_MAX_M=30000 SF_TILE_SHAPE=2 python tools/tile_lab.py 25088
} > $OUT/r03_l_tile_lab.txt 2>&1
grep -v amdgpu.ids $OUT/r03_l_tile_lab.txt
