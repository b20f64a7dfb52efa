# fused heads kernel with the action loops unrolled (A = 12 at compile time) vs the runtime-A form (DTC_HEADS_UNROLL=0): digests, times,
# tests, interleaved bench
T=deep-tracking-control_amd/tools
for rep in 1 2; do
python $T/heads_hash.py 2>&1 | grep -v amdgpu | sed "s/^/unrolled: /"
DTC_HEADS_UNROLL=0 python $T/heads_hash.py 2>&1 | grep -v amdgpu | sed "s/^/runtime A: /"
done
timeout 1800 python -m pytest tests/test_hip_ppo.py tests/test_hip_h2i.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench unrolled', round(d['ms_per_step'],3), round(d['value']))"
DTC_HEADS_UNROLL=0 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench runtime-A', round(d['ms_per_step'],3), round(d['value']))"
done
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
