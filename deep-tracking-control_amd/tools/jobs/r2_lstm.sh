# recurrent paths (LSTM / multi-layer) + a bench line -> gpurun_out/r2/
mkdir -p gpurun_out/r2
timeout 1800 python -m pytest tests/test_hip_lstm.py tests/test_lstm_path.py tests/test_hip_gru.py tests/test_gru_path.py tests/test_composite_path.py -m gpu -x -q 2>&1 | tail -25 | tee gpurun_out/r2/lstm_tests.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('VALUE', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'], d['roofline']['frac'])
"
