# weight image A/B: accuracy tests, kernel timing, bench
for w in 1 0; do
echo "== DTC_S3_WIMG=$w"
DTC_S3_WIMG=$w timeout 900 python -m pytest tests/test_hip_split.py tests/test_hip_kernels.py -m gpu -q -x 2>&1 | tail -3
DTC_S3_WIMG=$w python deep-tracking-control_amd/tools/s3_ablate.py "wimg=$w" | tail -1
done
for w in 1 0 1 0; do
DTC_S3_WIMG=$w timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench wimg=$w', round(d['ms_per_step'],3), round(d['value']), d['gemm_accuracy']['split_bf16x3'])"
done
