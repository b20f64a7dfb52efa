O=gpurun_out; mkdir -p $O
python -m pytest tests/test_hip_h2i.py tests/test_hip_ppo.py tests/test_composite_path.py -x -q 2>&1 | tail -5
for v in 1 0; do
DTC_H2I=$v python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('H2I=$v', round(d['ms_per_step'],2), round(d['value']))"
done
for b in 768 1024; do
DTC_WGRAD_H2I_BLOCKS=$b python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgrad blocks $b', round(d['ms_per_step'],2), round(d['value']))"
done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/r5_bench_insitu.json 2>$O/r5_bench_insitu.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r5_bench_insitu.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['dtype'])
for k,v in (d.get('gemm_accuracy_in_situ') or {}).items():
    if k!='measure': print(k, v)
print(d['gemm_accuracy'])
PY
tail -3 $O/r5_bench_insitu.err
