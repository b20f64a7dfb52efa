"""Where does the split-precision forward kernel's time go?  Same launch (24576 x 512 x 512) with operands that (a) stream from HBM / L2 as in
the trainer, (b) come from an 8-row source through the row gather (every load hits the CU's L1): the difference is what the
memory path costs; what remains is issue / LDS / barrier time inside the CU."""
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
big = torch.randn(M, K, device=DEV)
small = torch.randn(8, K, device=DEV)
idx_small = torch.randint(0, 8, (M,), device=DEV)
idx_perm = torch.randperm(M, device=DEV)
cases = {"stream (plain rows)": _ffi.segmat([_ffi.seg(big, 0, K)]),
         "gather, random rows of the same matrix": _ffi.segmat([_ffi.seg(big, 0, K, gather=True)], idx_perm),
         "gather from 8 rows (L1-resident X)": _ffi.segmat([_ffi.seg(small, 0, K, gather=True)], idx_small)}
for split in (False, True):
    for name, X in cases.items():
        for _ in range(3):
            ops.linear_fwd(X, W, b, Y, "relu", M=M, split=split)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            ops.linear_fwd(X, W, b, Y, "relu", M=M, split=split)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 50.0
        print(f"split={split} {name}: {us:.1f} us = {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s")
