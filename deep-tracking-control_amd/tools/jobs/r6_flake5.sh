#!/bin/bash
# round 6, fifth pass: what does the run-to-run difference of the 2-rank gloo runs need?  control (plain dp) first and last; two processes
# that share the device without any collective; dp with every kernel / copy serialised; dp with copies on blit kernels instead of SDMA;
# dp with kernel arguments in host memory.
cd "$(dirname "$0")/../.." || exit 1
out=../gpurun_out/r06_flake5.txt
: > $out
n=${1:-5}
run() {   # label, mode, env...
  label=$1; mode=$2; shift 2
  echo "== $label" >> $out
  env DTC_HEADS_UNROLL=1 "$@" timeout 1500 python tools/flake_probe.py $mode $n 2>&1 | grep -E "DIFFERS|SUMMARY|Error|error" | cut -c1-260 >> $out
}
run "control: dp" dp
run "two processes, no collective" pair
run "dp, AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3" dp AMD_SERIALIZE_KERNEL=3 AMD_SERIALIZE_COPY=3
run "dp, HSA_ENABLE_SDMA=0" dp HSA_ENABLE_SDMA=0
run "dp, HIP_FORCE_DEV_KERNARG=0" dp HIP_FORCE_DEV_KERNARG=0
run "dp, DTC_LANE_PRIO=none" dp DTC_LANE_PRIO=none
run "control again: dp" dp
cat $out
