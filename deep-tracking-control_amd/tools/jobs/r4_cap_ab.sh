for i in 1 2; do for cap in 24 1000 48; do for w in gru composite; do
DTC_WGRAD_SPLIT_CAP=$cap python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('cap $cap $w', round(d['ms_per_step'],2), round(d['value']))"
done; done; done
