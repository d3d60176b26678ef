#!/bin/bash
# builds and runs tools/gemm_bench.cpp on the GPU box (development tool)
set -e
cd "$(dirname "$0")/.."
P=markov-process-analysis-on-point-cloud_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I include tools/gemm_bench.cpp -L $P -lmpa_hip -Wl,-rpath,$PWD/$P -o gpurun_out/gemm_bench
./gpurun_out/gemm_bench
