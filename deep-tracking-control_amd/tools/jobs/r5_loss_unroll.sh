# unfused ppo_loss_kernel with the action loops unrolled (recurrent trainers): tests + interleaved composite / gru bench
timeout 2400 python -m pytest tests/test_hip_ppo.py tests/test_hip_kernels.py tests/test_gru_path.py tests/test_composite_path.py tests/test_hip_dp.py -x -q -m gpu 2>&1 | tail -2
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value']))"; }
for rep in 1 2; do
for w in composite gru; do
timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w unrolled"
DTC_HEADS_UNROLL=0 timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w runtime-A"
done
done
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
