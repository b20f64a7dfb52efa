# full GPU suite on the mid-round tree + weight-gradient slice sweep on the image kernels
O=gpurun_out/q5
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -12 > $O/pytest.log
tail -4 $O/pytest.log
for i in 1 2; do
for nb in 1536 1024 2048; do
DTC_WGRAD_H2I_BLOCKS=$nb timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>$O/b_${nb}_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('wgrad_blocks=$nb', d['value'], d['ms_per_step'], d['kernel_classes'].get('wimage'))"
done
done
find gpurun_out -type f -size +4M -delete
tail -qn 2 $O/*.err | sort | uniq -c | cut -c1-200
