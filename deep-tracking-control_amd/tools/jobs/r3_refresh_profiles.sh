# Round-3 profile set -> gpurun_out/r3p/ (converted into profiles/r03_* by tools/analysis/collect_profiles.py):
#   default bench line (split-precision GEMMs; incl. cpu_baseline, traffic and MFMA-busy PMC passes), the same with the single-pass fp32 MFMA
#   kernels, per-shape table, rocprofv3 kernel stats (serialised + overlapped), SQ counter tables of both modes, GRU / composite lines,
#   the accuracy log of the split kernels against fp64
O=gpurun_out/r3p
mkdir -p $O
R=$PWD
timeout 1500 python bench.py > $O/r03_bench_n1.json 2> $O/r03_bench_n1.err; tail -1 $O/r03_bench_n1.err; cut -c1-300 $O/r03_bench_n1.json
DTC_GEMM_SPLIT=0 timeout 900 python bench.py --no-cpu-baseline > $O/r03_bench_fp32mfma.json 2>/dev/null
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r03_bench_shapes.json 2>/dev/null
timeout 900 python deep-tracking-control_amd/tools/analysis/gemm_pmc.py collect $O/pmc_split > $O/r03_gemm_pmc.md 2>/dev/null
DTC_GEMM_SPLIT=0 timeout 900 python deep-tracking-control_amd/tools/analysis/gemm_pmc.py collect $O/pmc_fp32 > $O/r03_gemm_pmc_fp32mfma.md 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/rp_serial $R/$O/rp_overlap
DTC_OVERLAP_WGRAD=0 DTC_OVERLAP_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_serial -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/rp_serial.json 2> $R/$O/rp_serial.err
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_overlap -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/rp_overlap.json 2> $R/$O/rp_overlap.err
cd $R
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r03_bench_$w.json 2>/dev/null
done
timeout 600 python -m pytest tests/test_hip_split.py -m gpu -q -s 2>&1 | grep -E "err |TFLOP|passed|failed" > $O/r03_split_accuracy.log
python deep-tracking-control_amd/tools/soak.py 20 2>&1 | tail -1 > $O/r03_soak.log
timeout 900 python bench.py --cpu-baseline-full 2>/dev/null | tail -1 > $O/r03_cpu_baseline_full.json
rm -rf $O/pmc_split $O/pmc_fp32 $O/rp_serial/*/*trace* 2>/dev/null
find $O -name "*.csv" -size +2M -delete
ls $O
