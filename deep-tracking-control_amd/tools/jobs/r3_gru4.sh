# GRU step kernels: A/B of the product library against a variant (VARIANT=<tag>) on the two recurrent workloads + unit tests
V=$PWD/deep-tracking-control_amd/tools/_bin/libdtc_hip_$VARIANT.so
timeout 900 python -m pytest tests/test_hip_gru.py -m gpu -q -x 2>&1 | tail -2
for i in 1 2; do
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w product', round(d['ms_per_step'],2), round(d['value']))"
DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w $VARIANT', round(d['ms_per_step'],2), round(d['value']))"
done
done
