"""Times the latency-oriented split kernel of csrc/gru_s3.hip (128 x 128 tiles, both operands three stages ahead, 72 KiB LDS: two
workgroups per CU) on a LARGE product next to the general split kernel (loads one stage ahead, three workgroups per CU):
dX[24576, 512] = dZ[24576, 1536] W[1536, 512]."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, ops  # noqa: E402

DEV = "cuda:0"
R, H = 24576, 512
lib = _ffi.lib()
f32 = torch.float32
W = torch.randn(3 * H, H, device=DEV) / 39.0
dgh = torch.randn(R, 3 * H, device=DEV)
img = torch.empty(int(lib.dtc_gru_s3_image_bytes(H)) // 8 + 1, dtype=torch.float64, device=DEV)
_ffi.check(lib.dtc_gru_s3_image(_ffi.cptr(W, f32), _ffi.ptr(img), H, 1, _ffi.stream()), "image")
dX = torch.empty(R, H, device=DEV)


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / reps)
    return best


fl = 2.0 * R * 3 * H * H
for nparts in (1, 3):
    part = torch.empty(nparts, R, H, device=DEV)
    us = timed(lambda: _ffi.check(lib.dtc_gru_dgrad_parts_s3(_ffi.cptr(dgh, f32), _ffi.ptr(img), _ffi.ptr(part), R * H, R, H, nparts,
                                                             _ffi.stream()), "parts"))
    print(f"deep kernel, {nparts} chunk(s): {us:.1f} us = {fl / us / 1e6:.1f} TFLOP/s fp32-equivalent")
us = timed(lambda: ops.linear_dgrad(dgh, W, dX, split=True))
print(f"general kernel (incl. its image launch): {us:.1f} us = {fl / us / 1e6:.1f} TFLOP/s fp32-equivalent")
ref = dgh.double() @ W.double()
part = torch.empty(1, R, H, device=DEV)
_ffi.check(lib.dtc_gru_dgrad_parts_s3(_ffi.cptr(dgh, f32), _ffi.ptr(img), _ffi.ptr(part), R * H, R, H, 1, _ffi.stream()), "parts")
ops.linear_dgrad(dgh, W, dX, split=True)
print("max error / max output: deep %.2e general %.2e" % (float((part[0].double() - ref).abs().max() / ref.abs().max()),
                                                          float((dX.double() - ref).abs().max() / ref.abs().max())))
