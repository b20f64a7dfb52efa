mkdir -p gpurun_out/r4f
O=gpurun_out/r4f
timeout 900 python -m pytest tests/test_hip_images.py -m gpu -x -q -s -k "weight_gradients" > $O/t1.log 2>&1; echo "wgrad images rc=$?"
grep -a "wgrad\|passed\|failed\|Error\|assert" $O/t1.log | tail -30
timeout 600 python deep-tracking-control_amd/tools/img_probe.py wgrad 2>/dev/null
DTC_I3_RT=3 timeout 900 python -m pytest tests/test_hip_images.py -m gpu -x -q > $O/t_rt3.log 2>&1; echo "images rt=3 rc=$?"; tail -n 3 $O/t_rt3.log
