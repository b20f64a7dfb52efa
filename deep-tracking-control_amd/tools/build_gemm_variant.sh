#!/bin/bash
# variant library of the GEMM file for A/B runs (DTC_LIB=...): libdtc_hip_g<tag>.so = product objects + gemm.hip built with extra flags
#   usage: build_gemm_variant.sh <tag> [extra hipcc flags...]
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tools/_bin /tmp/gv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -I ../include -I csrc "$@" -c csrc/gemm.hip -o /tmp/gv/gemm_$tag.o || exit 1
objs=$(ls build/*.o | grep -v "/gemm.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gv/gemm_$tag.o -o tools/_bin/libdtc_hip_g$tag.so && echo built tools/_bin/libdtc_hip_g$tag.so
