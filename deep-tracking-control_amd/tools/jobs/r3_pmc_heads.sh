# Round 3: SQ counter passes over a serialised step, fused-heads threads-per-row A/B, the new bench line fields
O=gpurun_out/r3
mkdir -p $O
timeout 1500 python deep-tracking-control_amd/tools/analysis/gemm_pmc.py collect $O/gemm_pmc > $O/gemm_pmc.md 2> $O/gemm_pmc.err
tail -3 $O/gemm_pmc.err
cat $O/gemm_pmc.md
timeout 600 python -m pytest tests/test_hip_ppo.py -m gpu -x -q -k "heads or teacher_forced_64" 2>&1 | tail -3
for i in 1 2 3; do
for v in 8 4; do
echo -n "DTC_HEADS_TPR=$v: "
DTC_HEADS_TPR=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), {k: round(v['ms'],2) for k,v in d['kernel_classes'].items() if 'heads' in k}, d['roofline_planner_4096']['avg_launch_us'])"
done
done | tee $O/ab_heads.log
