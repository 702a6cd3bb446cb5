set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{
SF_TILE_MAX_M=12544 python tools/tile_lab.py 3136 6272 12544
for s in 3 4 5; do SF_TILE_MAX_M=12544 SF_TILE_SHAPE=$s python tools/tile_lab.py 3136 6272; done
for s in 1 2; do SF_TILE_MAX_M=12544 SF_TILE_SHAPE=$s python tools/tile_lab.py 6272 12544; done
} > $OUT/r03_d_tile_lab.txt 2>&1
python -m pytest tests/test_hip_parity.py -q -m gpu --tb=short -x -k "linear or f2 or base_full or forward_small or shapes_f7 or uint8 or submodule" 2>&1 | tail -15 > $OUT/r03_d_tests.log
for B in 1 2 4; do python tools/fwd_time.py $B 16; done > $OUT/r03_d_fwd_time.txt 2>&1
SF_TILE_MAX_M=12544 python tools/fwd_time.py 4 16 >> $OUT/r03_d_fwd_time.txt 2>&1
grep -v amdgpu.ids $OUT/r03_d_tile_lab.txt; tail -5 $OUT/r03_d_tests.log; grep -v amdgpu.ids $OUT/r03_d_fwd_time.txt
