# HIP stream priorities of the second compute lane / the weight-gradient streams: two interleaved rounds
for r in 1 2; do
for v in "" "aux" "side" "aux,side"; do
echo -n "DTC_LANE_PRIO='$v': "
DTC_LANE_PRIO=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))"
done
done
