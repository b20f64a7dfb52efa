mkdir -p gpurun_out/r4k
O=gpurun_out/r4k
python - <<'PY'
import sys; sys.path.insert(0,'deep-tracking-control_amd')
from dtc_amd import ops
for i in range(2):
    print('sustained fp32-eq TFLOP/s: zero', round(ops.mfma_sustained('cuda:0', False),1), ' random', round(ops.mfma_sustained('cuda:0', True),1))
PY
timeout 900 python -m pytest tests/test_hip_ppo.py -m gpu -x -q -k "activation_images or teacher_forced_64" > $O/t1.log 2>&1; echo "ppo images rc=$?"; tail -n 3 $O/t1.log
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/bench.json 2> $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['ms_per_step'], d['roofline']['frac'], d['roofline'].get('sustained_mfma'), d['roofline'].get('frac_of_sustained_random'))"
