# round-3 extras on the final build: DP rehearsal through bench.py's own launcher (gloo, all ranks on the one GPU), the DP tests,
# the CPU port on the whole workload
O=gpurun_out/r3p
mkdir -p $O
rm -f $O/r03_dp_rehearsal.log
for n in 2 8; do
for w in decoder composite; do
echo "== rehearsal N=$n workload=$w (python bench.py --gpus $n; gloo, all ranks on cuda:0)" >> $O/r03_dp_rehearsal.log
DTC_BENCH_BACKEND=gloo DTC_BENCH_DEVICE=0 timeout 1200 python bench.py --gpus $n --steps 2 --warmup 1 --workload $w --no-traffic 2>$O/dp_${n}_$w.err | tail -1 | cut -c1-1600 >> $O/r03_dp_rehearsal.log
done
done
cut -c1-260 $O/r03_dp_rehearsal.log
timeout 1200 python -m pytest tests/test_hip_dp.py -m gpu -q 2>&1 | tail -2
timeout 900 python bench.py --cpu-baseline-full 2>/dev/null | tail -1 > $O/r03_cpu_baseline_full.json; cat $O/r03_cpu_baseline_full.json | cut -c1-400
