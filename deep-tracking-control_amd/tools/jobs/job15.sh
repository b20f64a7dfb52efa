mkdir -p gpurun_out
R=$PWD
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x 2>&1 | tail -2
timeout 300 python deep-tracking-control_amd/tools/microbench.py gemm 2>&1 | grep "M=" | head -6
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/rp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rp -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rp_bench.json 2> $R/gpurun_out/rp_bench.err
tail -1 $R/gpurun_out/rp_bench.json | cut -c1-300
ls $R/gpurun_out/rp
