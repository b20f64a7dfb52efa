# composite: padded gradient buffers zeroed once per update and slot (default) vs before every scatter (DTC_PAD_ZERO_ALWAYS=1)
timeout 1200 python -m pytest tests/test_composite_path.py tests/test_hip_dp_g7.py -x -q -m gpu 2>&1 | tail -2
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value']))"; }
for rep in 1 2 3; do
timeout 600 python bench.py --workload composite --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "composite once"
DTC_PAD_ZERO_ALWAYS=1 timeout 600 python bench.py --workload composite --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "composite always"
done
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
