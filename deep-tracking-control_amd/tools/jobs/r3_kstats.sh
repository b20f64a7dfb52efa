# per-shape table + rocprofv3 kernel stats of the serialised schedule -> gpurun_out/r3k/
O=gpurun_out/r3k
mkdir -p $O
R=$PWD
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/bench_shapes.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/rp_serial
DTC_OVERLAP_WGRAD=0 DTC_OVERLAP_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_serial -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/rp_serial.json 2> $R/$O/rp_serial.err
cd $R
rm -rf $O/rp_serial/*/*trace* 2>/dev/null
find $O -name "*.csv" -size +2M -delete
find $O -name "*kernel_stats.csv" | head -1 | xargs head -30
