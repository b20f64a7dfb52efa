"""Grouped weight gradients on operand images, alone on the chip: the three-buffer kernel against the four-buffer one (DTC_WGRAD_RING4=1),
interleaved, on groups shaped like the bench step's (61 / 63 / 70 tiles).  Prints us per launch (HIP events) and whether the two write the same bits.

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


for name, shapes in GROUPS.items():
    jobs = images(shapes)
    tiles = sum(-(-N // 128) * -(-K // 128) for N, K in shapes)
    ws = ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, M), DEV)
    flop = sum(2.0 * M * N * K for N, K in shapes)
    outs, t = {}, {"0": [], "1": []}
    for rnd in range(3):
        for v in ("0", "1"):
            os.environ["DTC_WGRAD_RING4"] = v
            t[v].append(timed(lambda: h2i.wgrad_group(jobs, M, ws)))
            if rnd == 0:
                outs[v] = [(j[2].clone(), j[4].clone()) for j in jobs]
    same = all(torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) for a, b in zip(outs["0"], outs["1"]))
    nan = any(torch.isnan(a[0]).any().item() for a in outs["1"])
    print(f"{name} ({tiles} tiles, {len(shapes)} layers): three buffers {min(t['0']):7.1f} us ({flop / min(t['0']) / 1e6:6.1f} TFLOP/s)  "
          f"four buffers {min(t['1']):7.1f} us ({flop / min(t['1']) / 1e6:6.1f} TFLOP/s)  rounds {[round(x, 1) for x in t['0']]} / {[round(x, 1) for x in t['1']]}  "
          f"same bits: {same}{' NaN!' if nan else ''}", flush=True)
os.environ.pop("DTC_WGRAD_RING4", None)
