#!/bin/bash
# round 6, sixth pass (needs a box on which the control differs): unrolled vs run-time-A kernels, an agent-scope acquire at the start of the
# heads kernel, a device-wide synchronise before / after it, default stream priorities, copies without SDMA.
cd "$(dirname "$0")/../.." || exit 1
out=../gpurun_out/r06_flake6.txt
: > $out
n=${1:-5}
run() {   # label, env...
  label=$1; shift
  echo "== $label" >> $out
  env DTC_HEADS_UNROLL=1 "$@" timeout 1500 python tools/flake_probe.py dp $n 2>&1 | grep -E "DIFFERS|SUMMARY|Error|error" | cut -c1-150 >> $out
}
run "control: dp"
if ! grep -q DIFFERS $out; then run "control 2: dp"; fi
if ! grep -q DIFFERS $out; then echo "QUIET BOX: control never differed, nothing to learn here" >> $out; cat $out; exit 0; fi
run "DTC_HEADS_UNROLL=0" DTC_HEADS_UNROLL=0
run "DTC_HEADS_ACQ=1 (acquire fence at the start of the heads kernel)" DTC_HEADS_ACQ=1
run "device synchronise BEFORE the heads launch" PROBE_SYNC_HEADS=before
run "device synchronise AFTER the heads launch" PROBE_SYNC_HEADS=after
run "DTC_LANE_PRIO=none" DTC_LANE_PRIO=none
run "HSA_ENABLE_SDMA=0" HSA_ENABLE_SDMA=0
run "control again: dp"
cat $out
