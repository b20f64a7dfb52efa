# Round-4 profile set -> gpurun_out/r4p/ (converted into profiles/r04_* by tools/analysis/collect_profiles.py gpurun_out/r4p r04):
#   default bench line (incl. cpu_baseline, traffic, MFMA-busy passes and the sustained-MFMA roofs), single-pass fp32 line, per-shape table,
#   rocprofv3 kernel stats (serialised + overlapped), SQ counter tables, GRU / composite lines, accuracy logs, the N = 2 rehearsal
#   (incl. configs4_composite and the fail-fast line), and the round's experiments: image chain A/B, K-loop ablation and counters of
#   the image-operand kernels, zero- vs random-operand run, LDS-DMA bandwidth, slice-fill A/B
O=gpurun_out/r4p
mkdir -p $O
R=$PWD
T=deep-tracking-control_amd/tools
timeout 1500 python bench.py > $O/r04_bench_n1.json 2> $O/r04_bench_n1.err; tail -1 $O/r04_bench_n1.err; cut -c1-300 $O/r04_bench_n1.json
DTC_GEMM_SPLIT=0 timeout 900 python bench.py --no-cpu-baseline > $O/r04_bench_fp32mfma.json 2>/dev/null
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
timeout 600 python -m pytest tests/test_hip_split.py tests/test_hip_images.py -m gpu -q -s 2>&1 | grep -E "err |TFLOP|passed|failed|image" > $O/r04_split_accuracy.log
python $T/soak.py 20 2>&1 | tail -1 > $O/r04_soak.log
# ---- N = 2 on the one GPU of the box (gloo): the launcher path, configs4_composite, collective check; and the fail-fast line
( DTC_BENCH_DEVICE=0 DTC_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-traffic 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('N=2 rehearsal (both ranks on cuda:0, gloo):', round(d['value']), 'env-steps/s,', round(d['ms_per_step'],1), 'ms/step; workload:', c['workload'][:60])
print('  collectives per step', c['collectives_per_step'], ' all-reduce bytes per step and rank', c['allreduce_bytes_per_step_per_rank'], ' rank ms', c['rank_ms_per_step'])
print('  configs4_composite:', json.dumps(d['configs4_composite']))" ; echo "--- python bench.py --gpus 2 on this 1-GPU box:"; ( time python bench.py --gpus 2 --steps 1 --warmup 0 ) 2>&1 | grep -v amdgpu ) > $O/r04_dp_rehearsal.log 2>&1
# ---- the round's experiments
( echo "== image chain in the trainers: DTC_IMAGES=0 (default) vs 1, three interleaved runs each"
for i in 1 2 3; do
  for im in 0 1; do DTC_IMAGES=$im python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernel_classes']
print('DTC_IMAGES=$im', round(d['ms_per_step'],2), 'ms/step', round(d['value']), 'env-steps/s;  serialised pass: fwd %.1f dgrad %.1f wgrad %.1f reduce %.2f ms' % (k['linear_fwd']['ms'],k['linear_dgrad']['ms'],k['linear_wgrad']['ms'],k['wgrad_reduce']['ms']))"; done
done
echo "== weight-gradient slices filling every workgroup slot (DTC_WGRAD_S3_FILL=1) vs the default 8 slices"
for i in 1 2; do
  for f in 0 1; do DTC_WGRAD_S3_FILL=$f python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('DTC_WGRAD_S3_FILL=$f', round(d['ms_per_step'],2), 'ms/step')"; done
done ) > $O/r04_images_ab.log 2>&1
( echo "== per-launch times, image-operand kernels next to the converting kernels (tools/img_probe.py)"; python $T/img_probe.py 2>/dev/null; python $T/img_probe.py wgrad 2>/dev/null
  echo "== zero vs random operands (tools/i3_power.py)"; python $T/i3_power.py 2>/dev/null
  echo "== LDS-DMA bandwidth into LDS, GEMM access pattern without MFMAs (tools/probes/dma_bw.hip)"; $T/_bin/dma_bw 2>/dev/null
  echo "== ds_read_b64_tr_b16 / LDS-DMA out-of-range semantics (tools/probes/tr16_dma.hip)"; $T/_bin/tr16_dma 2>/dev/null | head -8
  ) > $O/r04_image_kernels.log 2>&1
bash $T/jobs/r4_ablate_all.sh > /dev/null 2>&1          # -> $O/r04_i3_ablation.txt (variant libraries: tools/build_variant.sh, see the script)
cd /tmp
python $R/$T/analysis/pmc_any.py $R/$O/pmc_i3 linear_i3_kernel -- python $R/$T/i3_ablate.py product > $R/$O/r04_i3_pmc.txt 2>&1
python $R/$T/analysis/pmc_any.py $R/$O/pmc_w wgrad_ -- python $R/$T/img_probe.py wgrad > $R/$O/r04_wgrad_pmc.txt 2>&1
cd $R
timeout 900 python bench.py --cpu-baseline-full 2>/dev/null | tail -1 > $O/r04_cpu_baseline_full.json
rm -rf $O/pmc_split $O/pmc_i3 $O/pmc_w $O/rp_serial/*/*trace* 2>/dev/null
find $O -name "*.csv" -size +2M -delete
ls $O
