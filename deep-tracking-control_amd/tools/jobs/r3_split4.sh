O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_split.py -m gpu -q -s 2>&1 | grep -E "TFLOP|passed|failed|Error|assert" | tail -40 > $O/split4.log
cat $O/split4.log
DTC_GEMM_SPLIT=1 timeout 1200 python -m pytest tests/test_hip_ppo.py -m gpu -q -k "teacher_forced" 2>&1 | tail -8
timeout 1200 python -m pytest tests/test_hip_ppo.py -m gpu -q -k "teacher_forced" 2>&1 | tail -8
for v in 1; do
echo -n "DTC_GEMM_SPLIT=$v: "
DTC_GEMM_SPLIT=$v DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null > $O/bench_split_$v.json; python -c "import json,sys; d=json.load(open('$O/bench_split_$v.json')); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), d['last_update'][:3])"
done | tee $O/ab_split4.log
