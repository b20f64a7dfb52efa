mkdir -p gpurun_out/r3
timeout 1500 python -m pytest tests/test_gru_path.py tests/test_hip_gru.py tests/test_lstm_path.py tests/test_hip_lstm.py tests/test_composite_path.py tests/test_hip_kernels.py -m gpu -q 2>&1 | tail -4
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null > gpurun_out/r3/bench_$w.json; python -c "
import json; d=json.load(open('gpurun_out/r3/bench_$w.json')); print('$w', round(d['ms_per_step'],2), round(d['value']), round(d['roofline']['frac'],3)); print({k:(round(v['ms'],1), v['launches']) for k,v in sorted(d['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])[:6]})"
done
