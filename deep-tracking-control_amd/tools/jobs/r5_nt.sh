# image stores of the GEMM epilogue with the non-temporal hint (variant library tools/_bin/libdtc_hip_nt.so): kernel timing + interleaved bench A/B
O=gpurun_out/r5e
mkdir -p $O
R=$PWD
T=deep-tracking-control_amd/tools
V=$R/$T/_bin/libdtc_hip_${VARIANT:-nt}.so
{
for rep in 1 2 3; do
  python $T/h2i_probe.py all time 2>&1 | grep -v amdgpu | tr '\n' ' '; echo " | product"
  DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 python $T/h2i_probe.py all time 2>&1 | grep -v amdgpu | tr '\n' ' '; echo " | ${VARIANT:-nt}"
done
for rep in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench product', round(d['ms_per_step'],3), round(d['value']))"
DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench ${VARIANT:-nt}', round(d['ms_per_step'],3), round(d['value']))"
done
DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 timeout 900 python -m pytest tests/test_hip_h2i.py -x -q -m gpu 2>&1 | tail -2
} 2>&1 | tee $O/${VARIANT:-nt}_ab.txt
