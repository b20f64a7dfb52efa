mkdir -p gpurun_out profiles
R=$PWD
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/t15.log; cat gpurun_out/t15.log
timeout 300 python bench.py > gpurun_out/bench3.json 2> gpurun_out/bench3.err; tail -2 gpurun_out/bench3.err; cat gpurun_out/bench3.json
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_bench -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_bench.log 2>&1
tail -3 $R/gpurun_out/prof_bench.log
ls $R/gpurun_out/prof_bench
head -30 $R/gpurun_out/prof_bench/bench_kernel_stats.csv
