# A/B sweep of the switches that may have moved with the two-term fp16 kernels (same box, interleaved, bench --steps 10)
run() { env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-64s %.0f env-steps/s  %.2f ms' % ('$*', d['value'], d['ms_per_step']))"; }
for rep in 1 2; do
  run DTC_NOP=1
  run DTC_WGRAD_S3_BLOCKS=1024
  run DTC_WGRAD_S3_BLOCKS=2048 DTC_WGRAD_SPLIT_CAP=32
  run DTC_GEMM_SPLIT_MIN_RED=256
  run DTC_GEMM_SPLIT_MIN_COLS=256
  run DTC_GEMM_SPLIT_MIN_COLS=256 DTC_GEMM_SPLIT_MIN_RED=256
  run DTC_HEADS_TPR=8
done
