#!/bin/bash
# build the library here (hipcc cross-compiles gfx950), then run a command on a GPU box:  tools/gpu.sh [timeout_s] 'command'
set -e
cd "$(dirname "$0")/.."
python streamformer_amd/build.py > /tmp/sf_build.log 2>&1 || { grep -E "error" -A6 /tmp/sf_build.log | head -40; echo "BUILD FAILED"; exit 1; }
T=${1:-1200}; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
