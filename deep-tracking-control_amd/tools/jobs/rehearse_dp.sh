# N = 2 control-flow rehearsal of bench.py on a 1-GPU box: two ranks share cuda:0, collectives over gloo
# (RCCL refuses two ranks on one device).  Checks for deadlocks / ordering problems of the data-parallel path,
# not for performance.
for w in decoder composite; do
DTC_BENCH_BACKEND=gloo DTC_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
    --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --workload $w 2>&1 | tail -1 | cut -c1-200
done
