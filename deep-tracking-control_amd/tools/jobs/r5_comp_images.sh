# RecurrentDecoderPPO's policy step on operand images: parity tests + the composite workload A/B against the converting kernels
O=gpurun_out/q3
mkdir -p $O
timeout 1200 python -m pytest tests/test_composite_path.py tests/test_gru_path.py tests/test_hip_dp_g7.py -m gpu -x -q 2>&1 | tail -25 > $O/pytest.log
tail -25 $O/pytest.log
for i in 1 2; do
timeout 600 python bench.py --workload composite --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>$O/comp_img_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('images', d['value'], d['ms_per_step'])"
DTC_H2I=0 timeout 600 python bench.py --workload composite --steps 5 --warmup 2 --no-cpu-baseline --no-traffic 2>$O/comp_conv_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('converting', d['value'], d['ms_per_step'])"
done
DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload composite --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/comp_shapes.json 2> $O/comp_shapes.err
find gpurun_out -type f -size +4M -delete
tail -n 3 $O/*.err | cut -c1-300
