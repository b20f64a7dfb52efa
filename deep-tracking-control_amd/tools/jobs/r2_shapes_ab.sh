# per-shape serialised times with and without one environment setting:  r2_shapes_ab.sh "VAR=value" [filter]
mkdir -p gpurun_out/r2
S="$1"; F="${2:-linear_}"
for tag in base alt; do
  if [ $tag = base ]; then E="DTC_NOOP=1"; else E="$S"; fi
  env $E DTC_PROF_SHAPES=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_classes']
print('$tag', 'value %.0f ms %.2f roof %.1f' % (d['value'], d['ms_per_step'], d['roofline']['achieved']))
for n in sorted(k, key=lambda n:-k[n]['ms'])[:40]:
    if '$F' in n and k[n]['ms'] > 2.0: print('   %-34s %7.3f ms  %6.1f' % (n, k[n]['ms'], k[n]['rate']))
"
done | tee gpurun_out/r2/shapes_ab.txt
