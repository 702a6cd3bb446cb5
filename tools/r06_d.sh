#!/bin/bash
# round 6, call D: the fused streamed temporal projection + attention: parity (bit-identity with the two launches, oracle tests of the
# streamed path), alternating latency runs, timeline
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r06; mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -q -m gpu -x -k "stream or cache or tower or slide or several" > $OUT/d_tests.log 2>&1; tail -5 $OUT/d_tests.log
for i in 1 2 3; do
  echo -n "fused:   "; python tools/stream_trace.py 2>/dev/null
  echo -n "unfused: "; SF_DISABLE_STREAM_QKV_FUSE=1 python tools/stream_trace.py 2>/dev/null
done | tee $OUT/d_stream_ab.txt
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > /dev/null 2>&1
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S | cut -c1-150 | tee $OUT/d_streaming_timeline.txt
