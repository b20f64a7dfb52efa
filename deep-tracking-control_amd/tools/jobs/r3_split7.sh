O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_split.py -m gpu -q -s 2>&1 | grep -E "split=|passed|failed|Error|assert" | tail -8
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $O/suite7.log; tail -6 $O/suite7.log
for i in 1 2; do
for v in 1 0; do
echo -n "DTC_PACK_INPUTS=$v: "
DTC_PACK_INPUTS=$v DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null > $O/bench_pack_$v.json; python -c "import json,sys; d=json.load(open('$O/bench_pack_$v.json')); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), d['gemm_accuracy']['split_bf16x3'])"
done
done | tee $O/ab_pack.log
DTC_GEMM_SPLIT=0 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('fp32', round(d['ms_per_step'],3), round(d['value']))"
