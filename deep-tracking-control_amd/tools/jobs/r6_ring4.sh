# four-buffer grouped weight-gradient kernel (DTC_WGRAD_RING4=1) against the three-buffer one: alone on the chip, under the parity tests, in the step
O=gpurun_out; mkdir -p $O
timeout 300 python deep-tracking-control_amd/tools/wgrad_probe.py > $O/r06_ring4.txt 2>&1
timeout 300 python deep-tracking-control_amd/tools/wgrad_probe.py heavy >> $O/r06_ring4.txt 2>&1
DTC_WGRAD_RING4=1 timeout 600 python -m pytest tests/test_hip_h2i.py -x -q -m gpu -k "wgrad" 2>&1 | tail -3 >> $O/r06_ring4.txt
for r in 1 2 3; do for v in 0 1; do
DTC_WGRAD_RING4=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --detail $O/r6_ring4_detail.json 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ring4=$v', d['ms_per_step'], d['value'])" >> $O/r06_ring4.txt
done; done
cat $O/r06_ring4.txt
