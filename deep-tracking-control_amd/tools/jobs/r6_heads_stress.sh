#!/bin/bash
# round 6: is the unrolled fused heads + loss kernel itself non-deterministic (same inputs, repeated launches, other work on the GPU)?
cd "$(dirname "$0")/../.." || exit 1
out=../gpurun_out/r06_heads_stress.txt
: > $out
for u in 1 0; do
  for extra in "" "--noise 1" "--noise 1 --streams"; do
    NOISE_SECONDS=200 DTC_HEADS_UNROLL=$u timeout 600 python tools/heads_stress.py 384 ${1:-6000} $extra 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
  done
done
NOISE_SECONDS=200 DTC_HEADS_UNROLL=1 timeout 600 python tools/heads_stress.py 24576 2000 --noise 1 2>&1 | grep -v amdgpu.ids | tail -2 >> $out
cat $out
