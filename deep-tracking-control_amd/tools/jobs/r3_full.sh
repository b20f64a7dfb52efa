# round-end rehearsal: whole -m gpu suite on the default (split-precision) path, the PPO / kernel / recurrent suites on the single-pass path, smoke()
O=gpurun_out/r3
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/full_default.log; tail -3 $O/full_default.log
DTC_GEMM_SPLIT=0 timeout 2400 python -m pytest tests/test_hip_ppo.py tests/test_hip_kernels.py tests/test_composite_path.py tests/test_gru_path.py tests/test_hip_dp.py -m gpu -q 2>&1 | tail -6 > $O/full_fp32.log; tail -3 $O/full_fp32.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
