mkdir -p gpurun_out
R=$PWD
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -3
timeout 900 python bench.py > gpurun_out/r01_bench_n1.json 2> gpurun_out/r01_bench_n1.err; tail -2 gpurun_out/r01_bench_n1.err
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/rp_serial $R/gpurun_out/rp_overlap
DTC_OVERLAP_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rp_serial -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rp_serial.json 2> $R/gpurun_out/rp_serial.err
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/rp_overlap -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rp_overlap.json 2> $R/gpurun_out/rp_overlap.err
ls $R/gpurun_out/rp_serial $R/gpurun_out/rp_overlap
python - <<PY
import json
for f in ('r01_bench_n1','rp_serial','rp_overlap'):
    d=json.load(open('$R/gpurun_out/'+f+'.json'))
    print(f, d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_launch_us'], d.get('cpu_baseline',{}).get('value'))
PY
cd $R
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/r01_bench_$w.json 2>/dev/null
python -c "import json; d=json.load(open('gpurun_out/r01_bench_$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['achieved'])"
done
