#!/bin/bash
# variant libraries of the planner for tools/jobs/r2_planner_sweep.sh: libdtc_hip_fh<tag>.so = product objects + foothold.hip built with extra flags
#   usage: build_planner_variants.sh <tag> [extra hipcc flags...]
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tools/_bin /tmp/fhv
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -I ../include -I csrc "$@" -c csrc/foothold.hip -o /tmp/fhv/foothold_$tag.o || exit 1
objs=$(ls build/*.o | grep -v foothold.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/fhv/foothold_$tag.o -o tools/_bin/libdtc_hip_fh$tag.so && echo built tools/_bin/libdtc_hip_fh$tag.so
