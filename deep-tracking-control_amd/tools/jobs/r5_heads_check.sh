timeout 1500 python -m pytest tests/test_hip_h2i.py tests/test_hip_ppo.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), round(d['value']))"; done
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
