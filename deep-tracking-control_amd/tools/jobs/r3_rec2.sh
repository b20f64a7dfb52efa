# recurrent workloads: bench lines + per-shape tables -> gpurun_out/r3k/
O=gpurun_out/r3k
mkdir -p $O
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_$w.json 2>$O/bench_$w.err
python -c "import json; d=json.load(open('$O/bench_$w.json')); print('$w', round(d['ms_per_step'],2), round(d['value']))"
DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/bench_${w}_shapes.json 2>/dev/null
done
