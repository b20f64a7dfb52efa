set -x
mkdir -p gpurun_out/r2
python deep-tracking-control_amd/tools/microbench.py gemm > gpurun_out/r2/mb0.log 2>&1
DTC_PROF_SHAPES=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r2/bench0_shapes.json 2> gpurun_out/r2/bench0_shapes.err
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r2/bench0.json 2> gpurun_out/r2/bench0.err
tail -c 600 gpurun_out/r2/bench0.json
