# quick iteration: kernel tests, PPO parity (fast subset), bench (interleaved pairs with DTC_H2I=0), per-shape table
O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_h2i.py tests/test_h2image_format.py -x -q 2>&1 | tail -3
timeout 1200 python -m pytest tests/test_hip_ppo.py -x -q -k "teacher_forced_64 or activation_images or overlapped or diverged or free_running" 2>&1 | tail -3
for i in 1 2 3; do
for v in 1 0; do
DTC_H2I=$v timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('H2I=$v', round(d['ms_per_step'],2), round(d['value']))"
done; done
DTC_PROF_SHAPES=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ > $O/r5_shapes.json 2>/dev/null
python deep-tracking-control_amd/tools/analysis/shapes.py $O/r5_shapes.json 0.4
