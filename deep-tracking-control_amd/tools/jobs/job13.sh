mkdir -p gpurun_out
timeout 300 python deep-tracking-control_amd/tools/microbench.py gemm 2>&1 | tail -25
timeout 900 python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "linear or gemm or wgrad or dgrad or seg" 2>&1 | tail -4
