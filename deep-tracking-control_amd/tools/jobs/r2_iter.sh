# iteration loop on the GPU box: selected tests + the bench line -> gpurun_out/r2/
mkdir -p gpurun_out/r2
T=${1:-iter}
timeout 1200 python -m pytest tests/test_hip_kernels.py tests/test_hip_ppo.py tests/test_composite_path.py tests/test_hip_dp.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2/${T}_tests.log
tail -5 gpurun_out/r2/${T}_tests.log
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/${T}_bench.json 2> gpurun_out/r2/${T}_bench.err
tail -3 gpurun_out/r2/${T}_bench.err
python - <<PY
import json
d=json.loads(open('gpurun_out/r2/${T}_bench.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'], d['roofline']['frac'])
PY
