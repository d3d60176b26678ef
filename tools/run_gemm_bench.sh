#!/bin/bash
# builds and runs tools/gemm_bench.cpp on the GPU box (development tool).  run_gemm_bench.sh [dir holding libmpa_hip.so]
set -e
cd "$(dirname "$0")/.."
P=${1:-markov-process-analysis-on-point-cloud_amd}
P=$(realpath $P)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I include tools/gemm_bench.cpp -L $P -lmpa_hip -Wl,-rpath,$P -o gpurun_out/gemm_bench
./gpurun_out/gemm_bench
