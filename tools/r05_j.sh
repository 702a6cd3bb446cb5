#!/bin/bash
# round 5: wider skinny tiles (32 x 64 / 32 x 96) for the folded consumers of a streamed frame — parity, alternating A/B against the
# 32 x 32 tiles (SF_SKINNY_NB=1), kernel table
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r05
mkdir -p $OUT
cd $R
python -m pytest tests -q -m gpu -k "stream or cache or op_linear" > $OUT/j_tests.log 2>&1; tail -4 $OUT/j_tests.log
cd /tmp && export TMPDIR=/tmp
for i in 1 2 3; do
  echo -n "default (qkv 32 x 64 on 16 waves, MLP-up 32 x 96 on 8): "; python $R/tools/stream_trace.py 2>/dev/null
  echo -n "SF_SKINNY_NB=4 (qkv 32 x 32):                           "; SF_SKINNY_NB=4 python $R/tools/stream_trace.py 2>/dev/null
  echo -n "SF_SKINNY_NB=5 (MLP-up 32 x 96 on 4 waves):             "; SF_SKINNY_NB=5 python $R/tools/stream_trace.py 2>/dev/null
  echo -n "SF_SKINNY_NB=1 (32 x 32 everywhere):                    "; SF_SKINNY_NB=1 python $R/tools/stream_trace.py 2>/dev/null
done | tee $OUT/j_skinny_nb_ab.txt
rm -rf /tmp/prof_s
rocprofv3 --kernel-trace -d /tmp/prof_s -o s -- python $R/tools/stream_trace.py > /dev/null 2>&1
S=$(find /tmp/prof_s -name "*.db" | head -1)
python $R/tools/stream_timeline.py $S | cut -c1-130 | tee $OUT/j_streaming_timeline.txt
