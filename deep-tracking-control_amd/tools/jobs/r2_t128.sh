# 128 x 128 tiles: GEMM-level and PPO-level parity, then interleaved A/B against 128 x 64 -> gpurun_out/r2/
mkdir -p gpurun_out/r2
timeout 1800 python -m pytest tests/test_hip_kernels.py tests/test_hip_ppo.py -m gpu -x -q -k "not scorer and not plan and not heights" 2>&1 | tail -8 | tee gpurun_out/r2/t128_tests.log
bash deep-tracking-control_amd/tools/jobs/r2_ab.sh DTC_GEMM_128_BLOCKS=0 3
DTC_PROF_SHAPES=1 timeout 300 python bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_classes']
print('value %.0f ms %.2f roof %.1f frac %.4f' % (d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac']))
for n in sorted(k, key=lambda n:-k[n]['ms'])[:16]:
    print('   %-34s %7.3f ms  %6.1f' % (n, k[n]['ms'], k[n]['rate']))
" | tee gpurun_out/r2/t128_shapes.txt
