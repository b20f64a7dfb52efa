# exponent-delta table reads by inline assembly (no s_waitcnt vmcnt(0) at block borders): kernel tests + interleaved A/B against a library
# built with -DDTC_H2I_DT_ASM=0 (dtc_amd/lib/libdtc_hip_alt.so, built by hand beside the product library)
O=gpurun_out/q7
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_h2i.py -m gpu -x -q 2>&1 | tail -3
for i in 1 2 3; do
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>$O/a_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('asm', d['value'], d['ms_per_step'], k['linear_fwd']['ms'], k['linear_dgrad']['ms'])"
DTC_LIB=$PWD/deep-tracking-control_amd/dtc_amd/lib/libdtc_hip_alt.so timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>$O/b_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); k=d['kernel_classes']; print('c++', d['value'], d['ms_per_step'], k['linear_fwd']['ms'], k['linear_dgrad']['ms'])"
done
find gpurun_out -type f -size +4M -delete
tail -qn 2 $O/*.err | sort | uniq -c | cut -c1-200
