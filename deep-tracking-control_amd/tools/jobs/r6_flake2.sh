#!/bin/bash
# round 6, second pass: which tensor of the fused heads + loss launch differs first (before = its inputs, after = its outputs)
cd "$(dirname "$0")/../.." || exit 1
mkdir -p ../gpurun_out
out=../gpurun_out/r06_flake2.txt
: > $out
echo "== dp deep, DTC_HEADS_UNROLL=1" >> $out
PROBE_DEEP=1 DTC_HEADS_UNROLL=1 timeout 1200 python tools/flake_probe.py dp ${1:-6} 2>&1 | grep -v amdgpu.ids | tail -40 >> $out


cat $out
