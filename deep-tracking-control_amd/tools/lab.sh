#!/bin/bash
# build + run the stand-alone GEMM lab on the GPU box: tools/lab.sh <log-name> [args]
set -e
cd "$(dirname "$0")"
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -Wno-unused-value gemm_lab.hip -o _bin/gemm_lab -ldl
cd ../..
name=$1; shift
/usr/local/graft/bin/gpurun --timeout 300 -- "mkdir -p gpurun_out/r2; timeout 280 deep-tracking-control_amd/tools/_bin/gemm_lab $* > gpurun_out/r2/$name.log 2>&1" 2>&1 | grep -E "charged|status"
