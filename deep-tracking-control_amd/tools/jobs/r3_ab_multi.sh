# interleaved comparison of the product library with several variant libraries: VARIANTS="a b c" REPS=n
B=$PWD/deep-tracking-control_amd/tools/_bin
for i in $(seq ${REPS:-2}); do
timeout 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('product', round(d['ms_per_step'],3))"
for v in $VARIANTS; do
DTC_LIB=$B/libdtc_hip_$v.so DTC_SKIP_ABI_CHECK=1 timeout 600 python bench.py --steps 15 --warmup 4 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],3))"
done
done
