mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -4 > gpurun_out/t16.log; cat gpurun_out/t16.log
timeout 100 python deep-tracking-control_amd/tools/microbench.py scorer 2>&1 | tail -3
DTC_PROF_SHAPES=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench4.json 2> gpurun_out/bench4.err; tail -2 gpurun_out/bench4.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench4.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'])
rows=sorted(d['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])
for k,v in rows[:45]: print(f"{k:45s} {v['ms']:8.3f} ms  {v['launches']:4d}  {v['rate']:7.2f}")
PY
