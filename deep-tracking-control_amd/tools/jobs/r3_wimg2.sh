# grouped weight images: tests, then bench A/B (DTC_WIMG_GROUP)
timeout 900 python -m pytest tests/test_hip_split.py tests/test_hip_kernels.py -m gpu -q -x 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_hip_ppo.py -m gpu -q -x 2>&1 | tail -5
for w in 1 0 1 0; do
DTC_WIMG_GROUP=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench group=$w', round(d['ms_per_step'],3), round(d['value']))"
done
