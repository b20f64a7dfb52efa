# env-variable sweep of bench.py:  r2_sweep.sh VAR v1 v2 ...   -> one line per value
mkdir -p gpurun_out/r2
V=$1; shift
for x in "$@"; do
  env $V=$x timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_classes']
print('$V=$x', 'value %.0f ms %.2f roof %.1f' % (d['value'], d['ms_per_step'], d['roofline']['achieved']), 'wgrad %.2f reduce %.2f' % (k['linear_wgrad']['ms'], k['wgrad_reduce']['ms']))
" | tee -a gpurun_out/r2/sweep_$V.log
done
