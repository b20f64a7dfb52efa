# GRU recurrence on the split path (csrc/gru_s3.hip): tests, bench A/B (DTC_GRU_S3), per-shape table
O=gpurun_out/r3k
mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_gru.py tests/test_gru_path.py tests/test_composite_path.py -m gpu -q -x 2>&1 | tail -4
for g in 1 0; do
for w in gru composite; do
DTC_GRU_S3=$g timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$w gru_s3=$g', round(d['ms_per_step'],2), round(d['value']))"
done
done
DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload gru --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/bench_gru_shapes_s3.json 2>/dev/null
