mkdir -p gpurun_out/r3
for st in 0 1 2; do echo "prio $st"; DTC_S3_PRIO=$st python deep-tracking-control_amd/tools/s3_probe.py 2>&1 | grep "split=True stream"; done | tee gpurun_out/r3/s3_probe.log
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null > gpurun_out/r3/bench_cur.json; python -c "import json,sys; d=json.load(open('gpurun_out/r3/bench_cur.json')); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4)); print({k:round(v['ms'],2) for k,v in d['kernel_classes'].items() if 'wgrad' in k})"
