mkdir -p gpurun_out
for w in gru composite; do
DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err; tail -2 gpurun_out/bench_$w.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_$w.json'))
print('$w value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'])
rows=sorted(d['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])
print('total', sum(v['ms'] for k,v in rows))
for k,v in rows[:12]: print(f"{k:40s} {v['ms']:8.3f} ms  {v['launches']:5d}  {v['ms']/v['launches']*1e3:7.1f} us {v['rate']:7.2f}")
PY
done
