"""Per-kernel-class table of a bench.py JSON line (DTC_PROF_SHAPES=1): python shapes.py <bench.json> [min_ms]"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
rows = sorted(((v['ms'], k, v['launches'], v['rate']) for k, v in d['kernel_classes'].items()), reverse=True)
tot = sum(r[0] for r in rows)
gemm = sum(r[0] for r in rows if r[1].startswith('linear_'))
red = sum(r[0] for r in rows if r[1].startswith('wgrad_reduce'))
for ms, k, n, rate in rows:
    if ms >= thr:
        print(f"{k:42s} {n:4d} {ms:8.3f} ms {ms / n * 1e3:8.1f} us  {rate:8.2f}")
print(f"total {tot:.2f} ms, GEMM {gemm:.2f} ms, reduce {red:.2f} ms, launches {sum(r[2] for r in rows)}; value {d['value']:.0f} ms/step {d['ms_per_step']:.2f} roof {d['roofline']['achieved']:.1f}")
