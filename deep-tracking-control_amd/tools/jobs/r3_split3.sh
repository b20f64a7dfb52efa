# Round 3: split path, second pass: register-trimmed forward kernel, fused MSE layer, row-alignment timing, whole GPU suite with the path on
O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_split.py -m gpu -q -s 2>&1 | grep -E "TFLOP|passed|failed|Error|assert" | tail -40 > $O/split3.log
cat $O/split3.log
DTC_GEMM_SPLIT=1 timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -60 > $O/split3_suite.log
grep -E "passed|failed|FAILED|assert |AssertionError" $O/split3_suite.log | head -40
for v in 0 1; do
echo -n "DTC_GEMM_SPLIT=$v: "
DTC_GEMM_SPLIT=$v DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null > $O/bench_split_$v.json; python -c "import json,sys; d=json.load(open('$O/bench_split_$v.json')); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), d['last_update'][:3])"
done | tee $O/ab_split3.log
