# Round 3, first GPU pass: the whole -m gpu suite, the default bench line, and the data-parallel rehearsals THROUGH
# bench.py's own launcher (python bench.py --gpus N; ranks share the one GPU of the box over gloo)
O=gpurun_out/r3
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > $O/tests_a.log
tail -3 $O/tests_a.log
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_a.json 2> $O/bench_a.err; tail -2 $O/bench_a.err; cut -c1-600 $O/bench_a.json
for n in 2 8; do
for w in decoder composite; do
echo "== rehearsal N=$n workload=$w (python bench.py --gpus $n; gloo, all ranks on cuda:0)" >> $O/dp_rehearsal.log
DTC_BENCH_BACKEND=gloo DTC_BENCH_DEVICE=0 timeout 1200 python bench.py --gpus $n --steps 2 --warmup 1 --workload $w --no-traffic 2>$O/dp_${n}_$w.err | tail -1 | cut -c1-1600 >> $O/dp_rehearsal.log
done
done
cat $O/dp_rehearsal.log | cut -c1-400
