mkdir -p gpurun_out/r4full
O=gpurun_out/r4full
timeout 3000 python -m pytest tests/ -x -q -m gpu > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -n 8 $O/gpu_tests.log
python __graft_entry__.py smoke > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -n 2 $O/smoke.log
