mkdir -p gpurun_out/r3
V=$PWD/deep-tracking-control_amd/tools/_bin/libdtc_hip_w3noilv.so
timeout 600 python -m pytest tests/test_hip_split.py -m gpu -q 2>&1 | tail -2
for i in 1 2; do
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgrad ilv', round(d['ms_per_step'],3), round(d['value']), {k:round(v['ms'],2) for k,v in d['kernel_classes'].items() if 'linear_wgrad' in k})"
DTC_LIB=$V DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wgrad no-ilv', round(d['ms_per_step'],3), round(d['value']), {k:round(v['ms'],2) for k,v in d['kernel_classes'].items() if 'linear_wgrad' in k})"
done
