#!/bin/bash
# round 6, fourth pass: ONE process on exactly the data of data-parallel rank 0 / rank 1 (of 2): repeated runs + poisoned memory
cd "$(dirname "$0")/../.." || exit 1
out=../gpurun_out/r06_flake4.txt
: > $out
for r in 1 0; do
  echo "== one process on the data of rank $r, DTC_HEADS_UNROLL=1" >> $out
  PROBE_AS_RANK=$r DTC_HEADS_UNROLL=1 timeout 1200 python tools/flake_probe.py asrank ${1:-8} 2>&1 | grep -v "amdgpu.ids" | tail -16 >> $out
done
cat $out
