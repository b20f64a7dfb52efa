#!/bin/bash
# round 6, ninth pass: the compute lanes as streams of the library's own (default from here on) against torch's pooled high-priority streams
# (DTC_LANE_POOL=1: until round 6), which torch.distributed's gloo work streams are drawn from as well.
cd "$(dirname "$0")/../.." || exit 1
out=../gpurun_out/r06_flake9.txt
: > $out
n=${1:-5}
run() {   # label, mode, env...
  label=$1; mode=$2; shift 2
  echo "== $label" >> $out
  env DTC_HEADS_UNROLL=1 "$@" timeout 1500 python tools/flake_probe.py $mode $n 2>&1 | grep -E "DIFFERS|SUMMARY|Error|error" | grep -v "rank 1" | cut -c1-110 >> $out
}
run "pooled lanes (until round 6)" dp DTC_LANE_POOL=1
if ! grep -q DIFFERS $out; then run "pooled lanes again" dp DTC_LANE_POOL=1; fi
if ! grep -q DIFFERS $out; then echo "QUIET BOX" >> $out; cat $out; exit 0; fi
run "own lanes" dp
run "pooled lanes, round 2" dp DTC_LANE_POOL=1
run "own lanes, round 2" dp
run "own lanes, round 3" dp
run "pooled lanes, round 3" dp DTC_LANE_POOL=1
run "own lanes, round 4" dp
cat $out
