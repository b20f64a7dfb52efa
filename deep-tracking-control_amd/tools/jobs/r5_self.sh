# 128-row tiles: fragments of the tile a wave fetched itself read above the stage barrier (product) vs the plain order (variant "plain")
O=gpurun_out/r5e
mkdir -p $O
R=$PWD
T=deep-tracking-control_amd/tools
V=$R/$T/_bin/libdtc_hip_plain.so
{
echo "digests (product, then plain): must agree line by line"
python $T/h2i_hash.py 2>&1 | grep -v amdgpu
DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 python $T/h2i_hash.py 2>&1 | grep -v amdgpu
timeout 900 python -m pytest tests/test_hip_h2i.py tests/test_h2image_format.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2 3; do
  python $T/h2i_probe.py all time 2>&1 | grep -v amdgpu | tr '\n' ' '; echo " | product (self)"
  DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 python $T/h2i_probe.py all time 2>&1 | grep -v amdgpu | tr '\n' ' '; echo " | plain"
done
for rep in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench product (self)', round(d['ms_per_step'],3), round(d['value']))"
DTC_LIB=$V DTC_SKIP_ABI_CHECK=1 timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench plain', round(d['ms_per_step'],3), round(d['value']))"
done
python $T/h2i_trace.py 512 512 2>&1 | grep -v amdgpu | cut -c1-400 | head -3
python $T/h2i_trace.py 512 512 2>&1 | grep "row-tile position"
} 2>&1 | tee $O/self_ab.txt
