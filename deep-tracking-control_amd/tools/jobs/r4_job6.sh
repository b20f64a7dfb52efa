mkdir -p gpurun_out/r4h
O=gpurun_out/r4h
timeout 900 python -m pytest tests/test_hip_images.py -m gpu -x -q > $O/t1.log 2>&1; echo "images rc=$?"; tail -n 15 $O/t1.log
timeout 900 python -m pytest tests/test_hip_split.py -m gpu -x -q > $O/t2.log 2>&1; echo "split rc=$?"; tail -n 3 $O/t2.log
