# GRU backward on the split path: tests + bench A/B over the chunk count
timeout 1500 python -m pytest tests/test_hip_gru.py tests/test_gru_path.py tests/test_composite_path.py -m gpu -q -x 2>&1 | tail -4
for p in 6 3 2; do
for w in gru; do
DTC_GRU_S3_PARTS=$p timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w parts=$p', round(d['ms_per_step'],2), round(d['value']))"
done
done
timeout 600 python bench.py --workload composite --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('composite', round(d['ms_per_step'],2), round(d['value']))"
