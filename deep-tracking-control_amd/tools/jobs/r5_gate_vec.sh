# gru_gate_bwd with four hidden units per thread (16-byte accesses): recurrence tests + interleaved A/B of the recurrent workloads
O=gpurun_out/q8
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_gru.py tests/test_gru_path.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2; do
for v in 1 0; do
DTC_GRU_GATE_VEC=$v timeout 600 python bench.py --workload gru --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>$O/g_${v}_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('gru vec=$v', d['value'], d['ms_per_step'], k['gru_gate_bwd'])"
done
done
for v in 1 0; do
DTC_GRU_GATE_VEC=$v timeout 600 python bench.py --workload composite --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>$O/c_${v}.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('composite vec=$v', d['value'], d['ms_per_step'], k['gru_gate_bwd'])"
done
find gpurun_out -type f -size +4M -delete
