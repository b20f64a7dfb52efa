# RecurrentPPO: the packed observation images of the second update (regression test), recurrent suites, gru bench with the padded
# gradient buffers zeroed once per update vs per mini-batch
timeout 1500 python -m pytest tests/test_gru_path.py tests/test_composite_path.py tests/test_hip_gru.py -x -q -m gpu 2>&1 | tail -3
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value']))"; }
for rep in 1 2; do
timeout 600 python bench.py --workload gru --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "gru once"
DTC_PAD_ZERO_ALWAYS=1 timeout 600 python bench.py --workload gru --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "gru always"
done
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
