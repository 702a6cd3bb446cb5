#!/bin/bash
# round 4, batch R: streamed frame with the K = 768 K-parallel residual GEMM's tile loops unrolled (compile-time count) against the
# run-time loops (SF_DISABLE_SKG_UNROLL=1): p50 per frame, alternating, then the streaming parity tests
mkdir -p gpurun_out
L=gpurun_out/r04_stream_skg_unroll_ab.txt
: > $L
for i in 1 2 3; do
  echo "off: $(SF_REPS=4 SF_DISABLE_SKG_UNROLL=1 timeout 300 python tools/stream_trace.py 2>&1 | grep p50)" >> $L
  echo "on:  $(SF_REPS=4 timeout 300 python tools/stream_trace.py 2>&1 | grep p50)" >> $L
done
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -x -q -k "streaming or several_streams or sliding or vision_tower or op_linear" 2>&1 | tail -2 >> $L
cat $L
