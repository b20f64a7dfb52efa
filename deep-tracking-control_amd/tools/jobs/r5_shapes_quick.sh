O=gpurun_out/q1
mkdir -p $O
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ > $O/shapes.json 2> $O/shapes.err
DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload gru --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/gru_shapes.json 2> $O/gru_shapes.err
DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload composite --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/comp_shapes.json 2> $O/comp_shapes.err
find gpurun_out -type f -size +4M -delete
tail -2 $O/*.err | cut -c1-300
