# lane priorities again, on round 6's kernels (DTC_LANE_PRIO: which lanes are high-priority streams)
O=gpurun_out; mkdir -p $O; : > $O/r06_prio.txt
for rnd in 1 2 3; do for v in aux,side aux side none; do
  DTC_LANE_PRIO=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --detail $O/r6_prio_detail.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', d['ms_per_step'])" >> $O/r06_prio.txt
done; done
cat $O/r06_prio.txt
