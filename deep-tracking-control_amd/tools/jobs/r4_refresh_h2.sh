# Round-4 FINAL profile set (two-term fp16 GEMM path = the default) -> gpurun_out/r4q/ (-> profiles/r04_* by
# tools/analysis/collect_profiles.py gpurun_out/r4q r04): default bench line (cpu_baseline, traffic, sustained-MFMA roofs), the same
# step on the bf16 x 3 kernels (DTC_GEMM_SPLIT=1) and on the single-pass fp32 kernels (=0) on the same box, per-shape table, rocprofv3
# kernel stats (serialised + overlapped), SQ counter table, GRU / composite lines, accuracy logs, N = 2 rehearsal, soak, power samples,
# energy per launch
O=gpurun_out/r4q
mkdir -p $O
R=$PWD
T=deep-tracking-control_amd/tools
timeout 1500 python bench.py > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err; tail -1 $O/r04_bench_n1.err; cut -c1-300 $O/r04_bench_n1.json
DTC_GEMM_SPLIT=1 timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/r04_bench_bf16x3.json 2>/dev/null
DTC_GEMM_SPLIT=0 timeout 900 python bench.py --no-cpu-baseline --no-traffic > $O/r04_bench_fp32mfma.json 2>/dev/null
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r04_bench_shapes.json 2>/dev/null
timeout 900 python $T/analysis/gemm_pmc.py collect $O/pmc_split > $O/r04_gemm_pmc.md 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/rp_serial $R/$O/rp_overlap
DTC_OVERLAP_WGRAD=0 DTC_OVERLAP_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_serial -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/rp_serial.json 2> $R/$O/rp_serial.err
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_overlap -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/rp_overlap.json 2> $R/$O/rp_overlap.err
cd $R
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r04_bench_$w.json 2>/dev/null
done
timeout 600 python -m pytest tests/test_hip_split.py tests/test_hip_h2.py -m gpu -q -s 2>&1 | grep -E "err |TFLOP|passed|failed" > $O/r04_split_accuracy.log
( DTC_GEMM_SPLIT=1 timeout 600 python -m pytest tests/test_hip_split.py -m gpu -q -s 2>&1 | grep -E "err |passed|failed" | sed "s/^/[DTC_GEMM_SPLIT=1, bf16 x 3] /" ) >> $O/r04_split_accuracy.log
python $T/soak.py 100 2>&1 | tail -1 > $O/r04_soak.log
( DTC_BENCH_DEVICE=0 DTC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('N=2 rehearsal (both ranks on cuda:0, gloo):', round(d['value']), 'env-steps/s,', round(d['ms_per_step'],1), 'ms/step; workload:', c['workload'][:60])
print('  collectives per step', c['collectives_per_step'], ' all-reduce bytes per step and rank', c['allreduce_bytes_per_step_per_rank'], ' rank ms', c['rank_ms_per_step'])
print('  configs4_composite:', json.dumps(d['configs4_composite']))" ; echo "--- python bench.py --gpus 2 on this 1-GPU box:"; ( time python bench.py --gpus 2 --steps 1 --warmup 0 ) 2>&1 | grep -v amdgpu ) > $O/r04_dp_rehearsal.log 2>&1
bash $T/jobs/r4_h2_power.sh > $O/r04_h2_power.txt 2>&1
python $T/energy_probe.py 2>&1 | grep -v amdgpu > $O/r04_energy.txt
( echo "== DTC_GEMM_SPLIT=1 (bf16 x 3)"; DTC_GEMM_SPLIT=1 python $T/energy_probe.py 2>&1 | grep -v amdgpu ) >> $O/r04_energy.txt
timeout 900 python bench.py --cpu-baseline-full 2>/dev/null | tail -1 > $O/r04_cpu_baseline_full.json
rm -rf $O/pmc_split $O/rp_serial/*/*trace* $O/rp_overlap/*/*trace* 2>/dev/null
find $O -name "*.csv" -size +2M -delete
ls $O
