# socket power / shader clock while the bench step runs on the two-term fp16 path and on the bf16 x 3 path; serialised per-kernel sums
O=${O:-gpurun_out/r4h2}
mkdir -p $O
: > $O/h2_power.txt
sample() {
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' | sed "s/^/$1: /" >> $O/h2_power.txt; echo >> $O/h2_power.txt
    sleep 0.5
  done
}
for mode in 2 1; do
  DTC_GEMM_SPLIT=$mode python bench.py --steps 300 --warmup 5 --no-cpu-baseline --no-traffic > $O/power_bench_$mode.json 2>/dev/null &
  P=$!; sleep 8; sample "bench step, DTC_GEMM_SPLIT=$mode" $P
  python -c "import json; d=json.loads(open('$O/power_bench_$mode.json').read().strip().splitlines()[-1]); print('DTC_GEMM_SPLIT=$mode', d['value'], d['ms_per_step'])" >> $O/h2_power.txt
done
grep -v amdgpu $O/h2_power.txt | sed "s/GPU\[0\]\t\t: //g; s/=* Power Consumption =*//" | tail -40
for mode in 2 1; do
  DTC_GEMM_SPLIT=$mode DTC_PROF_SHAPES=1 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = d['kernel_classes']
tot = sum(v['ms'] for v in k.values())
print('DTC_GEMM_SPLIT=$mode serialised sum', round(tot, 2), 'ms; overlapped step', round(d['ms_per_step'], 2))
for n, v in sorted(k.items(), key=lambda kv: -kv[1]['ms'])[:14]:
    print('   ', n, v['launches'], round(v['ms'], 3), 'ms', round(v['ms'] / v['launches'] * 1e3, 1), 'us')
"
done
