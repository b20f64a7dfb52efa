mkdir -p gpurun_out/r4c
O=gpurun_out/r4c
timeout 900 python -m pytest tests/test_hip_images.py -m gpu -x -q -s > $O/t1.log 2>&1; echo "images rc=$?"
timeout 600 python deep-tracking-control_amd/tools/img_probe.py > $O/probe.log 2>&1; echo "probe rc=$?"
tail -n 25 $O/t1.log; cat $O/probe.log
