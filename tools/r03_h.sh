set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd $R
tools/bin/mfma_shape_lab 20000 > $OUT/r03_h_mfma_shape_lab.txt 2>&1
cat $OUT/r03_h_mfma_shape_lab.txt
python -m pytest tests -q -m gpu --tb=short -x 2>&1 | tail -25 > $OUT/r03_h_tests.log
tail -8 $OUT/r03_h_tests.log
python bench.py > $OUT/r03_h_bench_line.json 2> $OUT/r03_h_bench_err.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03_h_bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step","e2e_mfma_frac")}, d.get("latency_b1"), d["streaming"].get("p50_ms"), d["streaming"].get("first_pass"), d.get("accurate_mode"), d["train_step"].get("ms_per_step"))
PY
