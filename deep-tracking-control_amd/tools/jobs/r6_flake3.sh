#!/bin/bash
# round 6, third pass: does any kernel of the trainer read memory / LDS / registers it never wrote?  One rank, identical inputs; torch.empty
# buffers and the CUs' LDS + VGPRs hold a different pattern in every run.  All runs must be bit-identical to "ref".
cd "$(dirname "$0")/../.." || exit 1
mkdir -p ../gpurun_out
out=../gpurun_out/r06_flake3.txt
: > $out
for u in 1 0; do
  echo "== poison, DTC_HEADS_UNROLL=$u" >> $out
  DTC_HEADS_UNROLL=$u timeout 1200 python tools/flake_probe.py poison 8 2>&1 | grep -v "amdgpu.ids" | tail -24 >> $out
done
cat $out
