# interleaved A/B of the product library against a variant library: VARIANT=<tag> (tools/_bin/libdtc_hip_<tag>.so), REPS (default 3)
V=$PWD/deep-tracking-control_amd/tools/_bin/libdtc_hip_$VARIANT.so
for i in $(seq ${REPS:-3}); do
python deep-tracking-control_amd/tools/s3_ablate.py "product" | tail -1
DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 python deep-tracking-control_amd/tools/s3_ablate.py "$VARIANT" | tail -1
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench product', round(d['ms_per_step'],3), round(d['value']))"
DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench $VARIANT', round(d['ms_per_step'],3), round(d['value']))"
done
