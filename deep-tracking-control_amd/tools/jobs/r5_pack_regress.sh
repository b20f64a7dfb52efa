timeout 1500 python -m pytest tests/test_hip_ppo.py -x -q -m gpu -k "every_update_packs or overlapped_schedule or fused_heads" 2>&1 | tail -3
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
