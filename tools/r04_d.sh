#!/bin/bash
# round 4, batch D: whole-forward hipGraph replay at small batches (one / two clips, short clips) against eager launches
mkdir -p gpurun_out
L=gpurun_out/r04_forward_graph_ab.txt
: > $L
for bt in "1 16" "2 16" "1 4" "1 8"; do
  for off in 0 1; do
    if [ $off = 1 ]; then export SF_DISABLE_FORWARD_GRAPH=1; else unset SF_DISABLE_FORWARD_GRAPH; fi
    echo -n "graph_off=$off  " >> $L; timeout 200 python tools/fwd_time.py $bt 2>/dev/null | tail -1 >> $L
    echo -n "graph_off=$off fp32-accurate  " >> $L; SF_MODE=fp32 timeout 200 python tools/fwd_time.py $bt 2>/dev/null | tail -1 >> $L
  done
done
unset SF_DISABLE_FORWARD_GRAPH
cat $L
