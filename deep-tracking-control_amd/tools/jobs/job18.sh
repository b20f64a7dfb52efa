mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_kernels.py tests/test_hip_gru.py tests/test_gru_path.py -m gpu -q -x 2>&1 | tail -3
DTC_PROF_SHAPES=1 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench5.json 2> gpurun_out/bench5.err; tail -2 gpurun_out/bench5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench5.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'])
rows=sorted(d['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])
tot=0
for k,v in rows:
    if k.startswith('wgrad_reduce'): tot+=v['ms']; print(f"{k:45s} {v['ms']:8.3f} ms  {v['launches']:4d}  {v['ms']/v['launches']*1e3:7.1f} us")
print('reduce total', tot)
PY
