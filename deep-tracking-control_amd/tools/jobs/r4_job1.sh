mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests/test_hip_dp_g7.py tests/test_export.py -m gpu -x -q -s > gpurun_out/r4a/t1.log 2>&1; echo "t1 rc=$?"
timeout 1800 python -m pytest tests/test_hip_ppo.py -m gpu -q -s -k "unforced or free_running_4096 or teacher_forced_4096 or teacher_forced_64 or strict" > gpurun_out/r4a/t2.log 2>&1; echo "t2 rc=$?"
( time python bench.py --gpus 2 --steps 2 --warmup 1 ) > gpurun_out/r4a/failfast.log 2>&1; echo "ff rc=$?"
DTC_BENCH_DEVICE=0 DTC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 2 --warmup 1 --no-traffic > gpurun_out/r4a/dp2.json 2> gpurun_out/r4a/dp2.err; echo "dp2 rc=$?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"
tail -3 gpurun_out/r4a/t1.log gpurun_out/r4a/t2.log gpurun_out/r4a/failfast.log
