"""Times the image-operand forward kernel (24576 x 512 x 512, ReLU, fp32 result) of the library DTC_LIB points at: one line per run.
Used with the -DDTC_I3_PROBE=<mask> variants of csrc/gemm_s3.hip (tools/jobs/r4_ablate.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import ops  # noqa: E402

DEV = "cuda:0"
M, N, K = 24576, 512, 512
W = torch.randn(N, K, device=DEV) / 22.0
b = torch.randn(N, device=DEV)
Y = torch.empty(M, N, device=DEV)
Ximg = ops.AImage.from_tensor(torch.randn(M, K, device=DEV))
Yimg = ops.AImage(M, N, DEV)
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
imgs = ops.WeightImages()
best = 1e9
for rnd in range(3):
    with imgs:
        fn = (lambda: ops.linear_fwd_img(Ximg, W, b, Y, None, "relu")) if mode == "fp32" else (lambda: ops.linear_fwd_img(Ximg, W, b, None, Yimg, "relu"))
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 50.0)
print(f"{sys.argv[1] if len(sys.argv) > 1 else ''} [{mode}]: {best:.1f} us = {2.0 * M * N * K / best / 1e6:.1f} TFLOP/s fp32-equivalent")
