mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ppo.py -m gpu -q -x 2>&1 | tail -3
for ov in 0 1; do
DTC_OVERLAP_WGRAD=$ov timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_ov$ov.json 2> gpurun_out/bench_ov$ov.err; tail -1 gpurun_out/bench_ov$ov.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_ov$ov.json'))
print('overlap=$ov value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'])
PY
done
