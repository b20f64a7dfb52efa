# the actor's and the critic's recurrence in one launch per time step (dtc_gru_fwd_multi / dtc_gru_bwd_multi): tests, then interleaved A/B
O=gpurun_out/r5e
mkdir -p $O
timeout 1200 python -m pytest tests/test_hip_gru.py tests/test_composite_path.py tests/test_gru_path.py -x -q -m gpu 2>&1 | tail -4 | tee $O/gru_multi.txt
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value']))"; }
for rep in 1 2 3; do
for w in composite gru; do
timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w multi"
DTC_GRU_MULTI=0 timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w separate"
done
done | tee -a $O/gru_multi.txt
python deep-tracking-control_amd/tools/gru_pair_probe.py 2>&1 | grep -v amdgpu.ids | tee -a $O/gru_multi.txt
