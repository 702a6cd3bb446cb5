#!/bin/bash
# round 5: single-query temporal attention with whole-line loads — streaming parity tests, alternating A/B against the
# one-key-per-lane kernel (SF_TEMPORAL_DECODE_LANE_KEY=1), kernel table of the streamed frame
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
python -m pytest tests -q -m gpu -k "stream or cache or decode" > $OUT/h_tests.log 2>&1; tail -4 $OUT/h_tests.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  echo -n "lines:    "; python $R/tools/stream_trace.py 2>/dev/null
  echo -n "lane-key: "; SF_TEMPORAL_DECODE_LANE_KEY=1 python $R/tools/stream_trace.py 2>/dev/null
done | tee $OUT/h_decode_ab.txt
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > /dev/null 2>&1
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S | cut -c1-120 | tee $OUT/h_streaming_timeline.txt
python $R/profiles/summarize.py $S | cut -c1-150 | head -12 | tee $OUT/h_streaming_kernel_stats.txt
