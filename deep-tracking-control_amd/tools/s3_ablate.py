"""Times the split-precision forward kernel (24576 x 512 x 512, ReLU) of the library DTC_LIB points at: one line per run.
Used with the -DDTC_S3_PROBE=<mask> variants of csrc/gemm_s3.hip (tools/jobs/r3_ablate.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, ops  # noqa: E402

DEV = "cuda:0"
M, N, K = 24576, 512, 512
W = torch.randn(N, K, device=DEV) / 22.0
b = torch.randn(N, device=DEV)
Y = torch.empty(M, N, device=DEV)
X = _ffi.segmat([_ffi.seg(torch.randn(M, K, device=DEV), 0, K)])
for _ in range(5):
    ops.linear_fwd(X, W, b, Y, "relu", M=M, split=True)
best = 1e9
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.linear_fwd(X, W, b, Y, "relu", M=M, split=True)
    e1.record()
    torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1) * 50.0)
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''}: {best:.1f} us = {2.0 * M * N * K / best / 1e6:.1f} TFLOP/s fp32-equivalent")
