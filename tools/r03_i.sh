set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pmc_b1f -o f -- python $R/tools/b1_trace.py 1 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pmc_b1w -o w -- python $R/tools/b1_trace.py 1 > /dev/null 2>&1
python $R/profiles/pmc_traffic.py /tmp/pmc_b1f /tmp/pmc_b1w > $OUT/r03_b1_pmc_traffic.json
python - <<'PY'
import json
d=json.load(open("/root/repo/gpurun_out/r03_b1_pmc_traffic.json"))
for k,v in d["kernels"].items():
    if "tile" in k or "attn" in k: print(k[:80], v)
PY
