#!/bin/bash
# build the library here (hipcc cross-compiles gfx950), then run a command on a GPU box:  tools/gpu.sh [timeout_s] 'command'
set -e
cd "$(dirname "$0")/.."
python streamformer_amd/build.py | grep -v hipcc || true
T=${1:-1200}; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
