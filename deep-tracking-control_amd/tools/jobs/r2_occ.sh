# launch-quantisation experiment: limit resident workgroups per CU for the fwd / dgrad / grouped-wgrad kernels; per-shape
# serialised times of one bench step -> gpurun_out/r2/occ_*.txt
mkdir -p gpurun_out/r2
run() {  # tag, env...
  tag=$1; shift
  env "$@" DTC_PROF_SHAPES=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_classes']
print('$tag', 'value %.0f ms %.2f roof %.1f' % (d['value'], d['ms_per_step'], d['roofline']['achieved']))
for n in sorted(k, key=lambda n:-k[n]['ms'])[:14]:
    print('   %-34s %7.3f ms' % (n, k[n]['ms']))
" | tee gpurun_out/r2/occ_$tag.txt
}
run base DTC_NOOP=1
run fwd3 DTC_GEMM_OCC_FWD=3
run fwd4 DTC_GEMM_OCC_FWD=4
run dgrad3 DTC_GEMM_OCC_DGRAD=3
run dgrad4 DTC_GEMM_OCC_DGRAD=4
run dgrad5 DTC_GEMM_OCC_DGRAD=5
run wgrad3 DTC_GEMM_OCC_WGRAD=3
run wgrad4 DTC_GEMM_OCC_WGRAD=4
