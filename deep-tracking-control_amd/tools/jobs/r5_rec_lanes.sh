# recurrent workloads: the amax-once-per-update fix, and how much the second compute lane buys them (DTC_OVERLAP_LANES=0: one lane)
O=gpurun_out/r5e
mkdir -p $O
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value']))"; }
for rep in 1 2; do
for w in composite gru; do
timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w default"
DTC_OVERLAP_LANES=0 timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w one-lane"
done
done | tee $O/rec_lanes.txt
timeout 900 python -m pytest tests/test_composite_path.py tests/test_gru_path.py -x -q -m gpu 2>&1 | tail -3 | tee -a $O/rec_lanes.txt
