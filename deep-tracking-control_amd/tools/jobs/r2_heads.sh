# fused heads + loss: parity tests, then interleaved A/B of the bench -> gpurun_out/r2/
mkdir -p gpurun_out/r2
timeout 1800 python -m pytest tests/test_hip_ppo.py tests/test_hip_dp.py -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/r2/heads_tests.log
bash deep-tracking-control_amd/tools/jobs/r2_ab.sh DTC_FUSE_HEADS=0 3
