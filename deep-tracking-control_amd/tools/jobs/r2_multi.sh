# several environment settings, one bench run each:  r2_multi.sh "A=1 B=2" "A=3" ...
mkdir -p gpurun_out/r2
for S in "$@"; do
  env $S timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-60s value %.0f ms %.2f' % ('$S', d['value'], d['ms_per_step']))
"
done | tee gpurun_out/r2/multi.txt
