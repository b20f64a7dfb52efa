O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_h2i.py -x -q 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_ppo.py -x -q -k "teacher_forced_64 or activation_images or overlapped or diverged" 2>&1 | tail -4
for i in 1 2; do
for v in "DTC_H2I=1" "DTC_H2I=0"; do
env $v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v', round(d['ms_per_step'],2), round(d['value']))"
done; done
DTC_PROF_SHAPES=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ > $O/r5_shapes.json 2>/dev/null
python deep-tracking-control_amd/tools/analysis/shapes.py $O/r5_shapes.json 0.5
