mkdir -p gpurun_out/r4d
O=gpurun_out/r4d
for rt in 1 3; do
  DTC_I3_RT=$rt timeout 900 python -m pytest tests/test_hip_images.py -m gpu -x -q > $O/t_rt$rt.log 2>&1; echo "images rt=$rt rc=$?"
  tail -n 3 $O/t_rt$rt.log
  DTC_I3_RT=$rt timeout 600 python deep-tracking-control_amd/tools/img_probe.py 2>/dev/null > $O/probe_rt$rt.log
done
paste -d'|' $O/probe_rt1.log $O/probe_rt3.log | cut -c1-75,115-160
