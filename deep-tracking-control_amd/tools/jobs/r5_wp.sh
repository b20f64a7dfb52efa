# h2i_wpack_kernel with 2 / 4 stage groups per block (512 / 1024 threads): byte-exact format tests under each variant, interleaved bench A/B,
# and the weight-image launch's own time from the serialised per-class profile
O=gpurun_out/r5e
mkdir -p $O
R=$PWD
T=deep-tracking-control_amd/tools
{
for v in wp2 wp4; do
DTC_LIB=$R/$T/_bin/libdtc_hip_$v.so DTC_SKIP_ABI_CHECK=1 timeout 900 python -m pytest tests/test_h2image_format.py tests/test_hip_h2i.py -x -q -m gpu 2>&1 | tail -1
done
cls() { python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_classes']['wimage']; print('$1', 'wimage', round(k['ms'],3), 'ms /', k['launches'], 'launches; step', round(d['ms_per_step'],2))"; }
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | cls product
for v in wp2 wp4; do
DTC_LIB=$R/$T/_bin/libdtc_hip_$v.so DTC_SKIP_ABI_CHECK=1 DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | cls $v
done
for rep in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench product', round(d['ms_per_step'],3), round(d['value']))"
for v in wp2 wp4; do
DTC_LIB=$R/$T/_bin/libdtc_hip_$v.so DTC_SKIP_ABI_CHECK=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench $v', round(d['ms_per_step'],3), round(d['value']))"
done
done
} 2>&1 | tee $O/wp_ab.txt
