# two-stages-ahead GEMM variants: parity tests, then per-shape times with and without them -> gpurun_out/r2/
mkdir -p gpurun_out/r2
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_ppo.py -m gpu -x -q 2>&1 | tail -6 | tee gpurun_out/r2/deep_tests.log
run() {
  tag=$1; shift
  env "$@" DTC_PROF_SHAPES=1 timeout 300 python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_classes']
print('$tag', 'value %.0f ms %.2f roof %.1f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']))
tot=0
for n in sorted(k, key=lambda n:-k[n]['ms']):
    if n.startswith('linear_') and k[n]['ms'] < 2.3 and 'wgrad' not in n:
        tot+=k[n]['ms']; print('   %-34s %7.3f ms' % (n, k[n]['ms']))
print('   narrow total %.3f ms' % tot)
" | tee gpurun_out/r2/deep_$tag.txt
}
run off DTC_GEMM_DEEP_BLOCKS=0
run on704 DTC_GEMM_DEEP_BLOCKS=704
run on1100 DTC_GEMM_DEEP_BLOCKS=1100
