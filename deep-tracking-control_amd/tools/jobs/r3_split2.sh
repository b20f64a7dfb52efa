# Round 3: split-precision path incl. the weight gradients: accuracy tests, the PPO / composite / GRU parity suites with the
# path on, A/B of the bench
O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_split.py -m gpu -q -s 2>&1 | grep -E "err |TFLOP|passed|failed|Error|assert" | tail -60 > $O/split2.log
cat $O/split2.log
DTC_GEMM_SPLIT=1 timeout 1500 python -m pytest tests/test_hip_ppo.py tests/test_composite_path.py -m gpu -q -x 2>&1 | tail -15 > $O/split2_ppo.log
cat $O/split2_ppo.log
for i in 1 2; do
for v in 0 1; do
echo -n "DTC_GEMM_SPLIT=$v: "
DTC_GEMM_SPLIT=$v DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null > $O/bench_split_$v.json; python -c "import json,sys; d=json.load(open('$O/bench_split_$v.json')); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), d['last_update'][:3])"
done
done | tee $O/ab_split2.log
