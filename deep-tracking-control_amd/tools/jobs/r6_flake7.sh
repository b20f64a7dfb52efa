#!/bin/bash
# round 6, seventh pass: is the 4-byte device-to-device copy_ behind the loss kernels (the KL into the gradient header, data parallel only) what the
# run-to-run differences need?  "memcpy" = the trainer until round 6; default = the finalize launch writes the header itself (no copy).
cd "$(dirname "$0")/../.." || exit 1
out=../gpurun_out/r06_flake7.txt
: > $out
n=${1:-5}
run() {   # label, env...
  label=$1; shift
  echo "== $label" >> $out
  env DTC_HEADS_UNROLL=1 "$@" timeout 1500 python tools/flake_probe.py dp $n 2>&1 | grep -E "DIFFERS|SUMMARY|Error|error" | cut -c1-150 >> $out
}
run "old form: KL -> header by copy_ (hipMemcpyAsync D2D)" PROBE_KLCOPY=memcpy
if ! grep -q DIFFERS $out; then run "old form again" PROBE_KLCOPY=memcpy; fi
if ! grep -q DIFFERS $out; then echo "QUIET BOX: the old form never differed, nothing to learn here" >> $out; cat $out; exit 0; fi
run "new form: the finalize launch writes the header (no copy)"
run "KL -> header by an elementwise kernel instead of a memcpy" PROBE_KLCOPY=kernel
run "new form, second round"
run "old form: copy_, second round" PROBE_KLCOPY=memcpy
run "new form, third round"
cat $out
