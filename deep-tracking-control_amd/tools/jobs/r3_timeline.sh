# kernel timeline (start/end per dispatch) of the overlapped bench step -> gpurun_out/r3k/timeline/
mkdir -p gpurun_out/r3k
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r3k/timeline
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r3k/timeline -o tl --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/gpurun_out/r3k/timeline.json 2> $R/gpurun_out/r3k/timeline.err
ls -la $R/gpurun_out/r3k/timeline
cd $R
python deep-tracking-control_amd/tools/analysis/timeline.py $(find gpurun_out/r3k/timeline -name "*kernel_trace.csv" | head -1) 1 | tee gpurun_out/r3k/timeline.txt
rm -rf gpurun_out/r3k/timeline
