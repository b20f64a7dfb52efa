mkdir -p gpurun_out/r4b
O=gpurun_out/r4b
timeout 900 python -m pytest tests/test_export.py -m gpu -x -q > $O/t1.log 2>&1; echo "export rc=$?"
timeout 1800 python -m pytest tests/test_hip_ppo.py -m gpu -q -s -k "unforced or free_running_4096 or teacher_forced_4096" > $O/t2.log 2>&1; echo "ppo rc=$?"
DTC_BENCH_DEVICE=0 DTC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-traffic > $O/dp2.json 2> $O/dp2.err; echo "dp2 rc=$?"
DTC_WGRAD_S3_FILL=1 timeout 900 python -m pytest tests/test_hip_split.py -m gpu -x -q > $O/t3.log 2>&1; echo "split(fill) rc=$?"
for i in 1 2 3; do
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/a$i.json 2> $O/a$i.err
  DTC_WGRAD_S3_FILL=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/b$i.json 2> $O/b$i.err
done
DTC_PROF_SHAPES=1 DTC_WGRAD_S3_FILL=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/shapes_fill.json 2> $O/shapes_fill.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4b/[ab]?.json')):
    d=json.load(open(f)); print(f, round(d['ms_per_step'],2), round(d['value']))
d=json.load(open('gpurun_out/r4b/shapes_fill.json'))
for k,v in sorted(d['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])[:12]: print(k,v)
PY
tail -n 3 $O/t1.log $O/t2.log $O/t3.log
