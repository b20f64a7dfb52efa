mkdir -p gpurun_out/r4i
O=gpurun_out/r4i
timeout 2400 python -m pytest tests/test_hip_ppo.py -m gpu -x -q > $O/t_ppo.log 2>&1; echo "ppo rc=$?"; tail -n 6 $O/t_ppo.log
timeout 1200 python -m pytest tests/test_composite_path.py tests/test_hip_dp_g7.py -m gpu -x -q > $O/t_comp.log 2>&1; echo "composite+g7 rc=$?"; tail -n 4 $O/t_comp.log
for i in 1 2 3; do
  DTC_IMAGES=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/a$i.json 2> $O/a$i.err
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/b$i.json 2> $O/b$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4i/[ab]?.json')):
    try:
        d=json.load(open(f)); print(f, round(d['ms_per_step'],2), round(d['value']), d['roofline']['frac'])
    except Exception as e: print(f, 'ERR', e)
PY
tail -n 5 $O/b1.err
