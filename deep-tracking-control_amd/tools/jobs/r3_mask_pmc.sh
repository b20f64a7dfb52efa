# Round 3: ReLU sign-record kernels (tests + interleaved A/B of the bench), then the SQ counter passes over a serialised step
O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_ppo.py -m gpu -x -q -k "relu_sign or linear or teacher_forced or overlapped or strict" 2>&1 | tail -6 > $O/tests_mask.log
cat $O/tests_mask.log
for i in 1 2 3; do
for v in 1 0; do
echo -n "DTC_RELU_MASK=$v: "
DTC_RELU_MASK=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4))"
done
done | tee $O/ab_mask.log
timeout 1500 python deep-tracking-control_amd/tools/analysis/gemm_pmc.py collect $O/gemm_pmc > $O/gemm_pmc.md 2> $O/gemm_pmc.err
tail -3 $O/gemm_pmc.err
cat $O/gemm_pmc.md
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_shapes.json 2>/dev/null
