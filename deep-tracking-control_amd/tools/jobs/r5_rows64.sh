# 64-row tiles for the image-operand launches with at most one 128 x 128 tile per CU: kernel tests on both tile heights + interleaved A/B
O=gpurun_out/q4
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_h2i.py tests/test_h2image_format.py -m gpu -x -q 2>&1 | tail -8 > $O/pytest.log
tail -8 $O/pytest.log
for i in 1 2 3; do
for mx in 256 0 384; do
DTC_H2I_ROWS64_MAX=$mx timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>$O/b_${mx}_$i.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('rows64_max=$mx', d['value'], d['ms_per_step'])"
done
done
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ > $O/shapes.json 2> $O/shapes.err
find gpurun_out -type f -size +4M -delete
tail -qn 2 $O/*.err | sort | uniq -c | cut -c1-200
