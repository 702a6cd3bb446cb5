#!/bin/bash
# build (here or on the GPU box) and run the GEMM lab
set -e
cd "$(dirname "$0")/.."
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/gemm_lab tools/gemm_lab.hip streamformer_amd/csrc/sf_gemm.hip streamformer_amd/csrc/sf_gemm256.hip streamformer_amd/csrc/sf_gemm_panel.hip
[ "$1" = "build" ] || ./tools/gemm_lab "$@"
