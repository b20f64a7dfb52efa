# N = 8 rehearsal of bench.py (all ranks on the one GPU of the box, gloo): control flow, collective-sequence check, DP fields
mkdir -p gpurun_out/r2
for w in decoder composite; do
echo "== rehearsal N=8 workload=$w (gloo, all ranks on cuda:0)"
DTC_BENCH_BACKEND=gloo DTC_BENCH_DEVICE=0 timeout 1500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 \
    --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 8 --steps 2 --warmup 1 --workload $w --no-traffic 2>&1 | tail -1 | cut -c1-1600
done | tee gpurun_out/r2/dp8_rehearsal.log
