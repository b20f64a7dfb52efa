mkdir -p gpurun_out/r3
for st in 0 1 2 3 5; do echo "stagger $st"; DTC_S3_STAGGER=$st python deep-tracking-control_amd/tools/s3_probe.py 2>&1 | grep "split=True stream"; done | tee gpurun_out/r3/s3_probe.log
