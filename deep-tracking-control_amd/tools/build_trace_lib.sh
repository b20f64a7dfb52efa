#!/bin/bash
# libdtc_hip_trace.so: the product library built with -DDTC_TRACE (per-block time stamps in the forward GEMM), for tools/gemm_lab
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DDTC_TRACE -fhip-fp32-correctly-rounded-divide-sqrt -I ../include -I csrc -shared csrc/gemm.hip csrc/runtime.hip -o tools/_bin/libdtc_hip_trace.so
