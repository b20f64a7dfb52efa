# Round-6 profile set (operand-image chain = the default) -> gpurun_out/r6q/ (-> profiles/r06_* by tools/analysis/collect_profiles.py
# gpurun_out/r6q r06): default bench line (cpu_baseline, traffic, in-situ accuracy, sustained-MFMA roofs), the same step on round 4's
# converting kernels (DTC_H2I=0) and on the single-pass fp32 kernels on the same box, per-shape table, rocprofv3 kernel stats (serialised +
# overlapped), SQ counter table of the GEMM family, planner counters, GRU / composite lines, per-workgroup timeline of the 512 x 512 layer
O=gpurun_out/r6q
mkdir -p $O
R=$PWD
T=deep-tracking-control_amd/tools
timeout 1800 python bench.py --detail $R/$O/r06_bench_n1_detail.json > $O/r06_bench_n1.json 2> $O/r06_bench_n1.err; tail -1 $O/r06_bench_n1.err; cut -c1-300 $O/r06_bench_n1.json
timeout 1800 python bench.py --gpus 1 --steps 20 --warmup 5 --detail $R/$O/r06_bench_driver_command_detail.json > $O/r06_bench_driver_command.json 2> $O/driver_command.err; cut -c1-200 $O/r06_bench_driver_command.json
DTC_H2I=0 timeout 900 python bench.py --no-cpu-baseline --no-traffic --detail $R/$O/r06_bench_converting_detail.json > $O/r06_bench_converting.json 2> $O/converting.err
DTC_GEMM_SPLIT=0 timeout 900 python bench.py --no-cpu-baseline --no-traffic --detail $R/$O/r06_bench_fp32mfma_detail.json > $O/r06_bench_fp32mfma.json 2> $O/fp32mfma.err
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ --detail $R/$O/r06_bench_shapes_detail.json > $O/r06_bench_shapes.json 2> $O/shapes.err
timeout 900 python $T/analysis/gemm_pmc.py collect $O/pmc_split > $O/r06_gemm_pmc.md 2> $O/gemm_pmc.err
python $T/h2i_trace.py 512 512 2>/dev/null > $O/r06_h2i_timeline.txt
python $T/h2i_trace.py 256 512 2>/dev/null >> $O/r06_h2i_timeline.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/rp_serial $R/$O/rp_overlap
DTC_OVERLAP_WGRAD=0 DTC_OVERLAP_LANES=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_serial -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-in-situ > $R/$O/rp_serial.json 2> $R/$O/rp_serial.err
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_overlap -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-in-situ > $R/$O/rp_overlap.json 2> $R/$O/rp_overlap.err
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
i=$((i+1))
rm -rf $R/$O/pmc_sc_$i
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/$O/pmc_sc_$i -o pmc --output-format csv -- python $R/$T/prof_target.py scorer > /dev/null 2>&1
done
rm -rf $R/$O/kt_sc4096
timeout 300 rocprofv3 --kernel-trace -d $R/$O/kt_sc4096 -o kt --output-format csv -- python $R/$T/prof_target.py scorer4096 > /dev/null 2>&1
cd $R
python - <<'PY' > gpurun_out/r6q/r06_planner_pmc.md
import csv, glob, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int); dur = []
for d in sorted(glob.glob('gpurun_out/r6q/pmc_sc_*')):
    for f in glob.glob(d + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'foothold_plan' in r['Kernel_Name']:
                tot[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']] += 1
    for f in glob.glob(d + '/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'foothold_plan' in r['Kernel_Name']:
                dur.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
print('# foothold_plan_fast_kernel, 98 304 maps per launch: rocprofv3 PMC (round 6; tools/jobs/r6_refresh_profiles.sh, prof_target.py scorer)\n')
print('| counter | per launch |\n|---|---|')
for k in sorted(tot):
    # SQ_* counters are summed over the dispatches of a pass; FETCH / WRITE in the guide's units (FETCH_SIZE x 2 for 16 B/lane loads on gfx950)
    per = tot[k] / max(1, len(dur) / 3)
    print(f'| {k} | {per:.4g} |')
if dur:
    print(f'\nkernel duration over {len(dur)} launches: mean {sum(dur)/len(dur)/1e3:.1f} us, min {min(dur)/1e3:.1f} us')
    f, w = tot.get('FETCH_SIZE', 0) / max(1, len(dur) / 3), tot.get('WRITE_SIZE', 0) / max(1, len(dur) / 3)
    print(f'HBM-side bytes per launch: FETCH_SIZE (KB) x 2 (gfx950 wide-load correction) + WRITE_SIZE (KB) = {(2 * f + w) * 1024 / 1e6:.1f} MB (counters in KB); algorithmic 3096 B x 98304 = 304.3 MB')
    v, wc = tot.get('SQ_ACTIVE_INST_VALU', 0), tot.get('SQ_BUSY_CYCLES', 0)
    d4 = []
    for f in glob.glob('gpurun_out/r6q/kt_sc4096/**/*kernel_trace.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'foothold_plan' in r['Kernel_Name']:
                d4.append(float(r['End_Timestamp']) - float(r['Start_Timestamp']))
    if d4:
        d4 = d4[5:] or d4
        m4 = sum(d4) / len(d4) / 1e3
        print(f'\nBASELINE configs[3], ONE launch over 4096 envs x 4 legs (12.68 MB algorithmic): rocprofv3 kernel-only duration over {len(d4)} launches: '
              f'mean {m4:.2f} us, min {min(d4)/1e3:.2f} us -> {3096.0 * 4096 / (m4 * 1e-6) / 1e9:.0f} GB/s = {3096.0 * 4096 / (m4 * 1e-6) / 1e9 / 8000:.3f} of 8 TB/s '
              f'(16 workgroups of 256 envs on 256 CUs: a latency-bound launch, not a bandwidth-bound one)')
    print(f'SQ_INSTS_VALU per env {tot.get("SQ_INSTS_VALU", 0) / max(1, len(dur) / 3) / 98304 * 64 / 64:.0f} wave-instructions x 1/64; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = {v / max(1.0, wc):.2f}')
PY
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --detail $R/$O/r06_bench_${w}_detail.json > $O/r06_bench_$w.json 2>/dev/null
done
timeout 600 python -m pytest tests/test_hip_h2i.py -m gpu -q -s 2>&1 | grep -E "err |timing|passed|failed" > $O/r06_h2i_accuracy.log
rm -rf $O/pmc_split $O/pmc_sc_* $O/kt_sc4096 $O/rp_serial/*/*trace* $O/rp_overlap/*/*trace* 2>/dev/null
find $O -type f -size +1M -delete
find gpurun_out -type f -size +4M -delete
du -sk gpurun_out/* | sort -n | tail -5
du -sk $O/* | sort -n | tail -8
for f in $O/*.err; do echo "== $f"; tail -2 $f | cut -c1-300; done
