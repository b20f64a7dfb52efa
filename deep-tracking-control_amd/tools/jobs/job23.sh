timeout 900 python -m pytest tests/test_hip_gru.py tests/test_gru_path.py tests/test_composite_path.py tests/test_abi_and_host.py -m gpu -q -x 2>&1 | tail -3
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])"
done
