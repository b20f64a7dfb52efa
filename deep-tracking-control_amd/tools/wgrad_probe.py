"""Grouped weight gradients on operand images, alone on the chip, on groups shaped like the bench step's (61 / 64 / 70 tiles): us per launch
(HIP events, kernel + reduce) and a checksum of the gradients' bits -- run it under DTC_LIB=<variant library> to compare builds
(tools/build_variant.sh, tools/jobs/r6_wgrad_var.sh).

    python deep-tracking-control_amd/tools/wgrad_probe.py [heavy]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import h2i, ops  # noqa: E402

DEV = "cuda:0"
M = 24576
heavy = "heavy" in sys.argv
GROUPS = {
    "61 tiles": [(512, 693), (512, 512), (512, 512), (128, 256), (256, 128), (12, 128)],
    "70 tiles": [(512, 693), (512, 512), (512, 512), (256, 512), (128, 256), (256, 128), (12, 128), (64, 128)],
    "64 tiles": [(512, 693), (512, 512), (512, 512), (256, 512)],
}


def images(shapes):
    jobs = []
    for N, K in shapes:
        dZ, X = torch.randn(M, N, device=DEV), torch.randn(M, K, device=DEV)
        if heavy:                                    # rows of very different magnitude: every block rescales
            dZ *= torch.exp2(torch.randint(-20, 20, (M, 1), device=DEV).float())
            X *= torch.exp2(torch.randint(-20, 20, (M, 1), device=DEV).float())
        jobs.append((h2i.HImage.from_tensor(dZ), h2i.HImage.from_tensor(X), torch.empty(N, K, device=DEV), 0, torch.empty(N, device=DEV)))
    return jobs


def timed(fn, n=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


torch.manual_seed(7)
for name, shapes in GROUPS.items():
    jobs = images(shapes)
    tiles = sum(-(-N // 128) * -(-K // 128) for N, K in shapes)
    ws = ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, M), DEV)
    flop = sum(2.0 * M * N * K for N, K in shapes)
    t = [timed(lambda: h2i.wgrad_group(jobs, M, ws)) for _ in range(3)]
    cw = sum(int(j[2].view(torch.int32).sum(dtype=torch.int64).item()) for j in jobs) & 0xffffffff
    cb = sum(int(j[4].view(torch.int32).sum(dtype=torch.int64).item()) for j in jobs) & 0xffffffff
    print(f"{name} ({tiles} tiles, {len(shapes)} layers): {min(t):7.1f} us ({flop / min(t) / 1e6:6.1f} TFLOP/s)  rounds {[round(x, 1) for x in t]}  "
          f"bits dW {cw:08x} db {cb:08x}", flush=True)
