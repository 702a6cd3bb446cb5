set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
python -m pytest tests/test_hip_parity.py -q -m gpu --tb=short -x -k "stream or tower or cache" 2>&1 | tail -12 > $OUT/r03_m_tests.log
tail -12 $OUT/r03_m_tests.log
{
python tools/stream_first_pass.py
SF_DISABLE_STREAM_FUSED=1 python tools/stream_first_pass.py
python tools/stream_batch.py
} > $OUT/r03_m_stream.txt 2>&1
grep -v amdgpu.ids $OUT/r03_m_stream.txt
