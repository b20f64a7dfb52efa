#!/bin/bash
# variant library for A/B runs (DTC_LIB=...): product objects with ONE source rebuilt with extra flags
#   usage: build_variant.sh <tag> <source.hip> [extra hipcc flags...]   ->  tools/_bin/libdtc_hip_<tag>.so
cd "$(dirname "$0")/.."
tag=$1; src=$2; shift; shift
mkdir -p tools/_bin /tmp/gv
base=$(basename $src .hip)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fhip-fp32-correctly-rounded-divide-sqrt -I ../include -I csrc "$@" -c csrc/$src -o /tmp/gv/${base}_$tag.o || exit 1
objs=$(ls build/*.o | grep -v "/$base.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/gv/${base}_$tag.o -o tools/_bin/libdtc_hip_$tag.so && echo built tools/_bin/libdtc_hip_$tag.so
