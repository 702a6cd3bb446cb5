#!/bin/bash
# The device-sharing experiments of DESIGN.md 4 ("Device sharing"; results in profiles/r05_device_sharing.txt), one gpurun call each:
#   bash tools/head_det.sh pairs [outdir]      two processes per run: full + full, the same without the side stream, forward-only beside
#                                              full, one process with kernels of its own second stream in flight, two forward-only
#   bash tools/head_det.sh classes [outdir]    one forward-only victim; the other process loops ONE kernel class of the backward at a time
#   bash tools/head_det.sh variants [outdir]   the same with lab variants of sf_temporal_attn_bwd_kernel (needs build.py --lab)
# SF_POOL_SHARE_CU=1 in the environment gives the pooling-head kernels their exact LDS sizes (the state the finding was made in).
what=${1:-pairs}; out=${2:-gpurun_out/head_det_$what}; mkdir -p $out
N=${SF_DET_N:-200}
run2() { # tag envA envB
  tag=$1; shift
  ( env $1 SF_DET_TAG=$tag.a timeout 200 python tools/head_det.py > $out/$tag.a.log 2>&1 ) & pa=$!
  ( env $2 SF_DET_TAG=$tag.b timeout 200 python tools/head_det.py > $out/$tag.b.log 2>&1 ) & pb=$!
  wait $pa $pb
  tail -n 1 $out/$tag.a.log $out/$tag.b.log
}
victim_and_noise() { # seconds of the victim, environment of the noise process
  ( SF_DET_N=1000000 SF_DET_SECONDS=$1 SF_DET_FWD_ONLY=1 SF_DET_TS=1 SF_DET_TAG=victim timeout $(($1 + 120)) python tools/head_det.py > $out/victim.log 2>&1 ) & pv=$!
  sleep 20
  env $2 timeout $(($1 + 100)) python tools/noise_ops.py > $out/noise.log 2>&1
  wait $pv
  python - "$out" <<'PY'
import re, sys
out = sys.argv[1]
v = open(out + "/victim.log").read()
ev = [float(m.group(1)) for m in re.finditer(r"pass \d+ t=([\d.]+): differing", v)]
hb = [(int(m.group(1)), float(m.group(2))) for m in re.finditer(r"heartbeat pass (\d+) t=([\d.]+)", v)]
for m in re.finditer(r"\[noise\] (\S+) from ([\d.]+) to ([\d.]+) \((\d+) calls\)", open(out + "/noise.log").read()):
    a, b = float(m.group(2)), float(m.group(3))
    n = sum(1 for t in ev if a <= t <= b)
    passes = [p for p, t in hb if a <= t <= b]
    npass = (max(passes) - min(passes)) if len(passes) > 1 else 0
    print(f"{m.group(1):22s} {n:5d} differing victim passes (~{npass} victim passes in the window), {m.group(4)} noise calls")
PY
}
case $what in
pairs)
  run2 X0 "SF_DET_N=$N" "SF_DET_N=$N"
  run2 X1 "SF_DET_N=$N SF_TRAIN_SIDE_STREAM=0" "SF_DET_N=$N SF_TRAIN_SIDE_STREAM=0"
  run2 X2 "SF_DET_N=$((N*2)) SF_DET_FWD_ONLY=1" "SF_DET_N=$N"
  env SF_DET_N=$N SF_DET_NOISE=24 SF_DET_TAG=X3 timeout 200 python tools/head_det.py > $out/X3.log 2>&1; tail -n 1 $out/X3.log
  run2 X4 "SF_DET_N=$((N*2)) SF_DET_FWD_ONLY=1" "SF_DET_N=$((N*2)) SF_DET_FWD_ONLY=1" ;;
classes)  victim_and_noise 250 "SF_NOISE_SECONDS=22" ;;
variants) victim_and_noise 150 "SF_LIB=lab SF_NOISE_SECONDS=15 SF_NOISE_MODES=idle,tbwd:0,tbwd:1,tbwd:2,tbwd:5,tbwd:3,tbwd:7,tbwd:6,tbwd:0" ;;
esac
