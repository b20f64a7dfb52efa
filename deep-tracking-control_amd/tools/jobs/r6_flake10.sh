#!/bin/bash
# round 6, tenth pass: the fused heads + loss kernel ALONE on fixed inputs (default-priority stream) while other PROCESSES launch bursts of short
# kernels on high-priority queues / on default-priority queues: does work of another process on a high-priority queue change its result?
cd "$(dirname "$0")/../.." || exit 1
out=../gpurun_out/r06_flake10.txt
: > $out
for u in 1 0; do
  for prio in high normal; do
    for heavy in 0 1; do
      NOISE_HEAVY=$heavy NOISE_PRIO=$prio NOISE_SECONDS=100 DTC_HEADS_UNROLL=$u timeout 600 python tools/heads_stress.py 384 ${1:-30000} --noise 2 --queues 4 2>&1 | grep -v amdgpu.ids | tail -1 | sed "s/^/noise prio=$prio heavy=$heavy: /" >> $out
    done
  done
done
cat $out
