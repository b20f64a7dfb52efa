# the bench step with variant libraries (tools/_bin/libdtc_hip_<tag>.so) against the product build, interleaved
O=gpurun_out; mkdir -p $O; : > $O/r06_step_ab.txt
B=deep-tracking-control_amd/tools/_bin
for rnd in 1 2 3; do for t in base $@; do
  lib=$B/libdtc_hip_$t.so; [ $t = base ] && lib=deep-tracking-control_amd/dtc_amd/lib/libdtc_hip.so
  DTC_LIB=$PWD/$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --detail $O/r6_ab_detail.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$t', d['ms_per_step'], d['value'])" >> $O/r06_step_ab.txt
done; done
cat $O/r06_step_ab.txt
