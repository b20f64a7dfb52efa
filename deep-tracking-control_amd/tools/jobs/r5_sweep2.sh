# fewer batch slices for the grouped weight gradients (less slab traffic) + the recurrent workloads with the shared r / z gate-block packs
O=gpurun_out/q6
mkdir -p $O
timeout 900 python -m pytest tests/test_gru_path.py tests/test_composite_path.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do
for nb in 1024 512 768; do
DTC_WGRAD_H2I_BLOCKS=$nb timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>$O/b_${nb}_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('wgrad_blocks=$nb', d['value'], d['ms_per_step'], k['linear_wgrad']['ms'], k['wgrad_reduce']['ms'])"
done
done
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>$O/$w.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$w', d['value'], d['ms_per_step'])"
done
find gpurun_out -type f -size +4M -delete
