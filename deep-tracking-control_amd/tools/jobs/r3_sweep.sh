run() { env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['value']))"; }
run A=0
run DTC_GEMM_SPLIT_MIN_COLS=128
run DTC_GEMM_SPLIT_MIN_RED=256
run DTC_WGRAD_S3_BLOCKS=1024
run DTC_WGRAD_S3_BLOCKS=3072
run A=0
run DTC_GEMM_SPLIT_MIN_COLS=128 DTC_GEMM_SPLIT_MIN_RED=128
