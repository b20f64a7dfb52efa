mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/t14.log; cat gpurun_out/t14.log
timeout 200 python deep-tracking-control_amd/tools/microbench.py gemm > gpurun_out/mb5.log 2>&1; cat gpurun_out/mb5.log
for A in 0 1 2 3 7; do DTC_GEMM_ABLATE=$A timeout 100 python deep-tracking-control_amd/tools/microbench.py ablate 2>&1 | grep ABLATE; done | tee gpurun_out/ablate2.log
