mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_ppo.py -m gpu -q -x 2>&1 | tail -3
for cfg in "0 0" "1 0" "1 1"; do
set -- $cfg
DTC_OVERLAP_WGRAD=$1 DTC_OVERLAP_LANES=$2 timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_ov.json 2> gpurun_out/bench_ov.err; tail -1 gpurun_out/bench_ov.err
python - <<PY
import json
d=json.load(open('gpurun_out/bench_ov.json'))
print('wgrad=$1 lanes=$2 value', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'], d['last_update'])
PY
done
