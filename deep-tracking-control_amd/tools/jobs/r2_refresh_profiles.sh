# Round-2 profile set -> gpurun_out/r2p/ (converted into profiles/r02_* by tools/analysis/collect_profiles.py):
#   bench line (default run), per-shape table, rocprofv3 kernel stats (serialised + overlapped), GRU / composite lines,
#   N = 2 / N = 4 data-parallel rehearsals of bench.py over gloo on one device (collective-sequence check inside bench.py)
O=gpurun_out/r2p
mkdir -p $O
R=$PWD
timeout 900 python bench.py > $O/r02_bench_n1.json 2> $O/r02_bench_n1.err; tail -1 $O/r02_bench_n1.err
DTC_PROF_SHAPES=1 timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r02_bench_shapes.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/rp_serial $R/$O/rp_overlap
DTC_OVERLAP_WGRAD=0 timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_serial -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/rp_serial.json 2> $R/$O/rp_serial.err
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/rp_overlap -o bench --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic > $R/$O/rp_overlap.json 2> $R/$O/rp_overlap.err
cd $R
for w in gru composite; do
timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic > $O/r02_bench_$w.json 2>/dev/null
done
for n in 2 4; do
for w in decoder composite; do
echo "== rehearsal N=$n workload=$w (gloo, all ranks on cuda:0)" >> $O/r02_dp_rehearsal.log
DTC_BENCH_BACKEND=gloo DTC_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n \
    --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 2 --warmup 1 --workload $w --no-traffic 2>&1 | tail -1 | cut -c1-1400 >> $O/r02_dp_rehearsal.log
done
done
python deep-tracking-control_amd/tools/soak.py 20 2>&1 | tail -1 > $O/r02_soak.log
python deep-tracking-control_amd/tools/rollout_profile.py 2>&1 | tail -1 >> $O/r02_soak.log
ls $O
