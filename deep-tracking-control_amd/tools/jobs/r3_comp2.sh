O=gpurun_out/r3
timeout 900 python -m pytest tests/test_composite_path.py -m gpu -q -s 2>&1 | grep -E "knife|compared|SKIPPED|own|passed|failed|Error|^E  " | head -20
timeout 1500 python -m pytest tests/test_hip_ppo.py -m gpu -q -s -k "4096 or teacher" 2>&1 | grep -E "^\[step|passed|failed|Error|^E  " | cut -c1-260 | tail -30
for b in 512 1280; do
DTC_WGRAD_S3_BLOCKS=$b timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$b', round(d['ms_per_step'],3), round(d['value']), {k:r.get(k) for k in ('achieved','frac','traffic','traffic_over_algorithmic','mfma_busy')})"
done
