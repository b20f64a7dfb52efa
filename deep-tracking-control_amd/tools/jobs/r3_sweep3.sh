# split-path thresholds (output columns / reduction length), two rounds
run() { env "$@" timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['value']))"; }
for r in 1 2; do
for c in "128 128" "256 256" "128 256" "256 128"; do set -- $c; run DTC_GEMM_SPLIT_MIN_COLS=$1 DTC_GEMM_SPLIT_MIN_RED=$2; done
done
