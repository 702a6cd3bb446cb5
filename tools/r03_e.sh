set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python tools/stream_first_pass.py > $OUT/r03_e_first_pass.txt 2>&1
SF_STREAM_GRAPH_PER_POSITION=1 python tools/stream_first_pass.py >> $OUT/r03_e_first_pass.txt 2>&1
SF_MODE=fp32 python tools/stream_first_pass.py >> $OUT/r03_e_first_pass.txt 2>&1
python tools/stream_batch.py > $OUT/r03_e_stream_batch.txt 2>&1
python -m pytest tests/test_hip_parity.py tests/test_tile_gemm.py -q -m gpu --tb=short -k "stream or cache or tower or tile or causal" 2>&1 | tail -15 > $OUT/r03_e_tests.log
python -m pytest tests/test_train_parity.py -q -m gpu --tb=long -k "rccl" 2>&1 | grep -E "mismatch|passed|failed" | cut -c1-1500 > $OUT/r03_e_rccl.log
grep -v amdgpu.ids $OUT/r03_e_first_pass.txt; grep -v amdgpu.ids $OUT/r03_e_stream_batch.txt | tail -8; tail -6 $OUT/r03_e_tests.log; cat $OUT/r03_e_rccl.log
