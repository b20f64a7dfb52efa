# env sweep on the final kernels (interleaved with the default)
run() { env "$@" timeout 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['value']))"; }
run A=0
run DTC_WGRAD_S3_BLOCKS=512
run DTC_WGRAD_S3_BLOCKS=1024
run DTC_WGRAD_S3_BLOCKS=1536
run A=0
run DTC_GEMM_SPLIT_MIN_COLS=64 DTC_GEMM_SPLIT_MIN_RED=64
run DTC_GEMM_SPLIT_MIN_COLS=256 DTC_GEMM_SPLIT_MIN_RED=128
run DTC_PACK_INPUTS=0
run DTC_FUSE_HEADS=0
run A=0
run DTC_RELU_MASK=0
run DTC_OVERLAP_LANES=0
run A=0
