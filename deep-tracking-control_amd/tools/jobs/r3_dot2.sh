B=$PWD/deep-tracking-control_amd/tools/_bin
timeout 600 python -m pytest tests/test_hip_split.py tests/test_hip_kernels.py -m gpu -q -s 2>&1 | grep -E "err |passed|failed|Error" | tail -14
python deep-tracking-control_amd/tools/s3_ablate.py "dot2 remainders" | tail -1
DTC_LIB=$B/libdtc_hip_nodot.so DTC_SKIP_ABI_CHECK=1 python deep-tracking-control_amd/tools/s3_ablate.py "unpack+sub remainders" | tail -1
python deep-tracking-control_amd/tools/s3_ablate.py "dot2 remainders" | tail -1
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), round(d['value']), d['gemm_accuracy']['split_bf16x3'])"
done
