# round-5 extras (one call, one box): rollout + update soak, the N = 2 rehearsal of bench.py (both ranks on cuda:0 over gloo) incl. the
# new N > 1 line fields, bench.py --gpus 2 on a 1-GPU box (must end with one JSON error line), the data-parallel GPU tests, three
# repeats of the driver's command (spread of one box), the CPU port on the WHOLE workload
O=gpurun_out/r5x
mkdir -p $O
T=deep-tracking-control_amd/tools
python $T/soak.py 100 2>&1 | tail -1 > $O/r05_soak.log; cat $O/r05_soak.log | cut -c1-300
( DTC_BENCH_DEVICE=0 DTC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('N=2 rehearsal (both ranks on cuda:0, gloo):', round(d['value']), 'env-steps/s,', round(d['ms_per_step'],1), 'ms/step; workload:', c['workload'][:60])
print('  collectives per step', c['collectives_per_step'], ' all-reduce bytes per step and rank', c['allreduce_bytes_per_step_per_rank'], ' rank ms', c['rank_ms_per_step'])
print('  communicator:', {k: c.get(k) for k in ('rccl_world', 'world', 'collective_sequence_ok')})
print('  configs4_composite:', json.dumps(d['configs4_composite']))" ; echo "--- python bench.py --gpus 2 on this 1-GPU box:"; ( time python bench.py --gpus 2 --steps 1 --warmup 0 ) 2>&1 | grep -v amdgpu ) > $O/r05_dp_rehearsal.log 2>&1
cat $O/r05_dp_rehearsal.log | cut -c1-400
timeout 1200 python -m pytest tests/test_hip_dp.py tests/test_hip_dp_g7.py -q -m gpu 2>&1 | tail -2 | tee -a $O/r05_dp_rehearsal.log
for i in 1 2 3; do
python bench.py --gpus 1 --steps 20 --warmup 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('driver command run $i:', round(d['ms_per_step'],2), 'ms', round(d['value']), 'env-steps/s, frac', round(d['roofline']['frac'],4), 'cpu', round(d['cpu_baseline']['value']))"
done | tee $O/r05_driver_repeats.txt
timeout 900 python bench.py --cpu-baseline-full 2>/dev/null | tail -1 > $O/r05_cpu_baseline_full.json; cut -c1-300 $O/r05_cpu_baseline_full.json
find gpurun_out -type f -size +4M -delete; rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
