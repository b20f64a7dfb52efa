# quick gate after a kernel change: split + kernel suites, forward kernel timing, two bench runs
timeout 900 python -m pytest tests/test_hip_split.py tests/test_hip_kernels.py -m gpu -q -x 2>&1 | tail -3
python deep-tracking-control_amd/tools/s3_ablate.py "fwd 24576x512x512" | tail -1
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), round(d['value']), d['gemm_accuracy']['split_bf16x3'])"
done
