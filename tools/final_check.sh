#!/bin/bash
# round-end check of HEAD on one GPU box: the whole GPU suite, then the default bench line
out=gpurun_out/final; mkdir -p $out
timeout 1000 python -m pytest tests -m gpu -q -x 2>&1 | tail -n 15 > $out/gpu_tests.log; tail -n 4 $out/gpu_tests.log
timeout 300 python bench.py > $out/bench.json 2> $out/bench.err; tail -c 600 $out/bench.json
