O=gpurun_out/r3k
mkdir -p $O
for p in 3 6; do
DTC_GRU_S3_PARTS=$p DTC_PROF_SHAPES=1 timeout 600 python bench.py --workload gru --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $O/bench_gru_shapes_p$p.json 2>/dev/null
done
