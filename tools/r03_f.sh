set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
{
python tools/stream_first_pass.py
SF_SKINNY_NT=1 python tools/stream_first_pass.py
python tools/tile_lab.py 196 1568
SF_SKINNY_NT=1 python tools/tile_lab.py 196 1568
} > $OUT/r03_f_stream.txt 2>&1
python -m pytest tests/test_hip_parity.py -q -m gpu --tb=short -k "stream or linear" 2>&1 | tail -5 > $OUT/r03_f_tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > $OUT/r03_f_streaming_run.txt 2>/dev/null
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S > $OUT/r03_f_streaming_timeline.txt
grep -v amdgpu.ids $OUT/r03_f_stream.txt; tail -3 $OUT/r03_f_tests.log; head -24 $OUT/r03_f_streaming_timeline.txt
