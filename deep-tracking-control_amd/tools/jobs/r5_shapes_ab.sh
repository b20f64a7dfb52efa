# per-shape HIP-event tables of one serialised step: operand-image chain (default) vs round 4's converting kernels (DTC_H2I=0)
O=gpurun_out; mkdir -p $O
for v in 1 0; do
DTC_H2I=$v DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r5_shapes_h2i$v.json 2>/dev/null
python deep-tracking-control_amd/tools/analysis/shapes.py $O/r5_shapes_h2i$v.json 0.25 > $O/r5_shapes_h2i$v.txt
done
paste -d'\n' /dev/null $O/r5_shapes_h2i1.txt; echo ----; cat $O/r5_shapes_h2i0.txt
