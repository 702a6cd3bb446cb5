set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{
python tools/stream_first_pass.py
SF_DISABLE_WEIGHT_PREFETCH=1 python tools/stream_first_pass.py
SF_SKINNY_NT=0 python tools/stream_first_pass.py
SF_MODE=fp32 python tools/stream_first_pass.py
} > $OUT/r03_g_stream.txt 2>&1
python -m pytest tests/test_hip_parity.py -q -m gpu --tb=short -k "stream or linear or tower" 2>&1 | tail -5 > $OUT/r03_g_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > $OUT/r03_g_streaming_run.txt 2>/dev/null
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S > $OUT/r03_g_streaming_timeline.txt
grep -v amdgpu.ids $OUT/r03_g_stream.txt; tail -3 $OUT/r03_g_tests.log; head -12 $OUT/r03_g_streaming_timeline.txt
