#!/bin/bash
# round 5: kernel tables of the forward (B = 8), one clip per call and the streamed frame with the folded pooling head
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_*
rocprofv3 --kernel-trace -d /tmp/prof_f -o f -- python $R/bench.py --profile --steps 12 --warmup 3 > $OUT/a_profile_run_line.json 2>/dev/null
python $R/profiles/summarize.py $(find /tmp/prof_f -name "*.db" | head -1) > $OUT/a_forward_kernel_stats.txt
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > $OUT/a_streaming_run.txt 2>/dev/null
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S > $OUT/a_streaming_timeline.txt
python $R/profiles/summarize.py $S > $OUT/a_streaming_kernel_stats.txt
rocprofv3 --kernel-trace -d /tmp/prof_b1 -o x -- python $R/tools/b1_trace.py 1 > /dev/null 2>&1
python $R/profiles/summarize.py $(find /tmp/prof_b1 -name "*.db" | head -1) > $OUT/a_b1_forward_kernel_stats.txt
grep -h "pool\|layernorm" $OUT/a_forward_kernel_stats.txt $OUT/a_streaming_kernel_stats.txt $OUT/a_b1_forward_kernel_stats.txt
