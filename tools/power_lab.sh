#!/bin/bash
# Lab (round 4): package power and shader clock (rocm-smi) sampled while the 8-clip bf16 forward runs back to back, and while idle
mkdir -p gpurun_out
L=gpurun_out/r04_power_lab.txt
: > $L
echo "== idle" >> $L
rocm-smi --showpower --showclocks --showmaxpower 2>/dev/null | grep -E "Power|sclk|mclk|Max" | head -8 >> $L
python - <<'PY' &
import sys, time, torch
sys.path.insert(0, ".")
import streamformer_amd as sa
cfg = sa.siglip_base()
m = sa.TimesformerMultiTaskingModelSigLIP(cfg); m.load_state_dict(sa.make_state_dict(cfg, 0)); m.to("cuda").eval()
x = torch.randn(8, 16, 3, 224, 224).cuda()
with torch.no_grad():
    t0 = time.time(); n = 0
    while time.time() - t0 < 24:
        for _ in range(50): m(x)
        torch.cuda.synchronize(); n += 50
    print(f"forwards: {n} in {time.time()-t0:.2f} s = {(time.time()-t0)/n*1e3:.3f} ms each", flush=True)
PY
PID=$!
sleep 18
for i in 1 2 3 4 5 6 7 8; do
  echo "== under load, sample $i" >> $L
  rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" | head -6 >> $L
  sleep 1
done
wait $PID >> $L 2>&1
cat $L
