mkdir -p gpurun_out
DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload composite --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_comp.json 2> gpurun_out/bench_comp.err; tail -2 gpurun_out/bench_comp.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_comp.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'])
rows=sorted(d['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])
print('total', sum(v['ms'] for k,v in rows))
for k,v in rows[:30]: print(f"{k:40s} {v['ms']:8.3f} ms  {v['launches']:5d}  {v['ms']/v['launches']*1e3:7.1f} us {v['rate']:7.2f}")
PY
