# A/B of one environment setting against the default, interleaved:  r2_ab.sh "VAR=value [VAR2=value2]" [repeats]
mkdir -p gpurun_out/r2
S="$1"; R=${2:-3}
for i in $(seq 1 $R); do
  for tag in base alt; do
    if [ $tag = base ]; then E="DTC_NOOP=1"; else E="$S"; fi
    env $E timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$tag', '$i', 'value %.0f ms %.2f' % (d['value'], d['ms_per_step']))
"
  done
done | tee gpurun_out/r2/ab.txt
