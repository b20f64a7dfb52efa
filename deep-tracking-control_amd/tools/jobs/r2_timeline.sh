# kernel timeline (start/end per dispatch) of the overlapped bench step -> gpurun_out/r2/timeline/
mkdir -p gpurun_out/r2
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/r2/timeline
timeout 600 rocprofv3 --kernel-trace -d $R/gpurun_out/r2/timeline -o tl --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > $R/gpurun_out/r2/timeline.json 2> $R/gpurun_out/r2/timeline.err
ls -la $R/gpurun_out/r2/timeline
cd $R
python deep-tracking-control_amd/tools/analysis/timeline.py $(find gpurun_out/r2/timeline -name "*kernel_trace.csv" | head -1) 1 | tee gpurun_out/r2/timeline.txt
