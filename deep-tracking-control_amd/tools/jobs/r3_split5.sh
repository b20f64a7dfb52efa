O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_split.py -m gpu -q -s 2>&1 | grep -E "split=|passed|failed|Error|assert" | tail -12 > $O/split5.log
cat $O/split5.log
DTC_GEMM_SPLIT=1 DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null > $O/bench_split_1.json; python -c "import json,sys; d=json.load(open('$O/bench_split_1.json')); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4))"
DTC_GEMM_SPLIT=1 timeout 1500 python deep-tracking-control_amd/tools/analysis/gemm_pmc.py collect $O/gemm_pmc_split > $O/gemm_pmc_split.md 2> $O/gemm_pmc_split.err
tail -3 $O/gemm_pmc_split.err
cat $O/gemm_pmc_split.md
