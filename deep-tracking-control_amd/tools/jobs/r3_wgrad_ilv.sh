# Round 3: interleaved-tile weight-gradient kernel (tests + interleaved A/B), device RNG tests, SQ counter passes
O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_rng.py tests/test_hip_ppo.py -m gpu -x -q -k "wgrad or rng or randperm or randn or update_draws or teacher_forced_64 or overlapped" 2>&1 | tail -6 > $O/tests_ilv.log
cat $O/tests_ilv.log
for i in 1 2 3; do
for v in 1 0; do
echo -n "DTC_WGRAD_ILV=$v: "
DTC_WGRAD_ILV=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), {k: round(v['ms'],2) for k,v in d['kernel_classes'].items() if 'wgrad' in k})"
done
done | tee $O/ab_ilv.log
timeout 1500 python deep-tracking-control_amd/tools/analysis/gemm_pmc.py collect $O/gemm_pmc > $O/gemm_pmc.md 2> $O/gemm_pmc.err
tail -3 $O/gemm_pmc.err
cat $O/gemm_pmc.md
