timeout 900 python -m pytest tests/test_hip_ppo.py tests/test_composite_path.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('decoder', d['value'], d['ms_per_step'], d['kernel_classes']['vae_loss'])"; done
