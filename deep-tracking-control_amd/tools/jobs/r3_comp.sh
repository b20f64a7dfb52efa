O=gpurun_out/r3
for e in "A=0" "DTC_GEMM_SPLIT_MIN_COLS=256 DTC_GEMM_SPLIT_MIN_RED=384" "DTC_GEMM_SPLIT=0"; do
echo "== $e"
env $e timeout 900 python -m pytest tests/test_composite_path.py -m gpu -q -s -k full_size 2>&1 | grep -E "knife|compared|SKIPPED|passed|failed|AssertionError|^E  .*assert" | head -12
done
bash deep-tracking-control_amd/tools/jobs/r3_sweep2.sh
