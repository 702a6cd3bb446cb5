set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{
for lab in 0 1 2 4 8 12 10; do SF_TILE_LAB=$lab python tools/tile_lab.py 3136; done
} > $OUT/r03_j_tile_lab.txt 2>&1
grep -v amdgpu.ids $OUT/r03_j_tile_lab.txt
