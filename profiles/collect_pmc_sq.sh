#!/bin/bash
# Clock under load and SQ activity per kernel, run ON the GPU box:  bash profiles/collect_pmc_sq.sh r02
# (PMC passes only: no trace domains beside them, one counter group per run)
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$R/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/px_*
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d /tmp/px_a -o a -- python $R/bench.py --profile --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/px_b -o b -- python $R/bench.py --profile --steps 2 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY --output-format csv -d /tmp/px_c -o c -- python $R/bench.py --profile --steps 2 --warmup 1 > /dev/null 2>&1
python $R/profiles/pmc_extra.py /tmp/px_a /tmp/px_b /tmp/px_c > $OUT/${TAG}_pmc_sq.json
head -c 3000 $OUT/${TAG}_pmc_sq.json
