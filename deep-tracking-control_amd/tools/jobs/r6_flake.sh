#!/bin/bash
# round 6: root cause of the data-parallel bit-for-bit flake (tests/test_hip_dp.py:113).  Separates: run-to-run differences of ONE
# rank with the overlapped schedule (no exchange at all), bucketed vs joined exchange over 2 gloo ranks, and both under
# AMD_SERIALIZE_KERNEL=3 (a missing dependency disappears when every kernel waits for the previous one).
cd "$(dirname "$0")/../.." || exit 1
mkdir -p ../gpurun_out
out=../gpurun_out/r06_flake.txt
: > $out
for u in 1 0; do
  echo "== solo, DTC_HEADS_UNROLL=$u" >> $out
  DTC_HEADS_UNROLL=$u timeout 900 python tools/flake_probe.py solo ${1:-8} 2>&1 | grep -v "^$" | tail -14 >> $out
done
echo "== dp, DTC_HEADS_UNROLL=1" >> $out
DTC_HEADS_UNROLL=1 timeout 900 python tools/flake_probe.py dp ${2:-5} 2>&1 | tail -24 >> $out
echo "== solo, DTC_HEADS_UNROLL=1 AMD_SERIALIZE_KERNEL=3" >> $out
AMD_SERIALIZE_KERNEL=3 DTC_HEADS_UNROLL=1 timeout 900 python tools/flake_probe.py solo ${1:-8} 2>&1 | tail -14 >> $out
cat $out
