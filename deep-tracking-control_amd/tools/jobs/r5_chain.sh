# narrow-layer chains (one launch per direction): kernel tests, trainer parity tests, interleaved A/B
O=gpurun_out/q9
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_h2i.py -m gpu -x -q 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_hip_ppo.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do
for c in 1 0; do
DTC_H2I_CHAIN=$c timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>$O/b_${c}_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('chain=$c', d['value'], d['ms_per_step'], sum(v['launches'] for v in k.values()), round(sum(v['ms'] for v in k.values()),2))"
done
done
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ > $O/shapes.json 2> $O/shapes.err
find gpurun_out -type f -size +4M -delete
tail -qn 3 $O/*.err | sort | uniq -c | cut -c1-300
