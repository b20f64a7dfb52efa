# how much of the chip one trainer leaves idle: two independent bench processes sharing the GPU vs one alone
mkdir -p gpurun_out/r2
one() { timeout 600 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 value %.0f ms %.2f' % (d['value'], d['ms_per_step']))"; }
one solo
one pairA & one pairB & wait
one solo2
