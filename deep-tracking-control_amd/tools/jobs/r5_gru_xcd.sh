# GRU time-step kernels: workgroup -> tile map by column tile per XCD (W_hh slice resident in the XCD's L2) vs by row tile
O=gpurun_out/r5e
mkdir -p $O
{
timeout 900 python -m pytest tests/test_hip_gru.py -x -q -m gpu 2>&1 | tail -2
for rep in 1 2; do
echo "colmap"; python deep-tracking-control_amd/tools/gru_pair_probe.py 2>&1 | grep -v amdgpu.ids
echo "rowmap"; DTC_GRU_XCD_COLS=0 python deep-tracking-control_amd/tools/gru_pair_probe.py 2>&1 | grep -v amdgpu.ids
done
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],2), 'ms', round(d['value']))"; }
for rep in 1 2; do
for w in composite gru; do
timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w colmap+multi"
DTC_GRU_MULTI=0 timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w colmap separate"
DTC_GRU_XCD_COLS=0 DTC_GRU_MULTI=0 timeout 600 python bench.py --workload $w --steps 4 --warmup 2 --no-cpu-baseline --no-traffic 2>/dev/null | line "$w rowmap separate"
done
done
} 2>&1 | tee $O/gru_xcd.txt
