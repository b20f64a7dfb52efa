mkdir -p gpurun_out
for P in 0 1 2; do for A in 0 1; do DTC_GEMM_PIPE=$P DTC_GEMM_ABLATE=$A timeout 100 python deep-tracking-control_amd/tools/microbench.py ablate 2>&1 | grep ABLATE | sed "s/^/PIPE=$P /"; done; done | tee gpurun_out/pipe1.log
