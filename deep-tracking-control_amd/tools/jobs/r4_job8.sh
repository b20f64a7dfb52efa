mkdir -p gpurun_out/r4j
O=gpurun_out/r4j
for i in 1 2 3; do
  DTC_IMAGES=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/a$i.json 2> $O/a$i.err
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/b$i.json 2> $O/b$i.err
  DTC_IMAGES=0 DTC_WGRAD_SPLIT_CAP=1000 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > $O/c$i.json 2> $O/c$i.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4j/[abc]?.json')):
    try:
        d=json.load(open(f)); k=d['kernel_classes']; print(f, round(d['ms_per_step'],2), round(d['value']), ' fwd %.1f dgrad %.1f wgrad %.1f reduce %.2f'%(k['linear_fwd']['ms'],k['linear_dgrad']['ms'],k['linear_wgrad']['ms'],k['wgrad_reduce']['ms']))
    except Exception as e: print(f, 'ERR', e)
PY
