"""Round-4 probe: ENERGY per launch of the step's kernel classes.  The bench step runs at the socket's power cap (~1380 W, rocm-smi), so
its time is its energy divided by that cap: this script loops one kernel for a few seconds, polls `rocm-smi --showpower` beside it and
reports mean power x time per launch = joules per launch (and per step, with the launch counts of one bench step)."""
import os
import re
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import h2i, ops  # noqa: E402

DEV = "cuda:0"


def power_now():
    out = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True).stdout
    p = re.search(r"Power \(W\): ([0-9.]+)", out)
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", out)
    return (float(p.group(1)) if p else float("nan")), (int(c.group(1)) if c else 0)


def measure(name, fn, per_step, seconds=4.0):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    samples, stop = [], threading.Event()

    def poll():
        time.sleep(1.0)
        while not stop.is_set():
            samples.append(power_now())
            time.sleep(0.3)
    th = threading.Thread(target=poll)
    th.start()
    n, t0 = 0, time.time()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.time() - t0 < seconds:
        for _ in range(50):
            fn()
        n += 50
        torch.cuda.synchronize()
    e1.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    us = e0.elapsed_time(e1) * 1e3 / n
    w = sum(s[0] for s in samples) / max(1, len(samples))
    mhz = sum(s[1] for s in samples) / max(1, len(samples))
    j = w * us * 1e-6
    print(f"{name:58s} {us:8.1f} us  {w:6.0f} W  {mhz:5.0f} MHz  {j * 1e3:7.2f} mJ / launch  x {per_step:4d} = {j * per_step:6.2f} J / step")
    return j * per_step


def main():
    g = torch.Generator(device=DEV).manual_seed(1)
    M = 24576
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)
    X, W, b, dZ = torch.relu(rn(M, 512)), rn(512, 512) / 22.0, rn(512), rn(M, 512) * 1e-3
    Y, dX = torch.empty(M, 512, device=DEV), torch.empty(M, 512, device=DEV)
    mask = ops.relu_mask(M, 512, DEV)
    Ximg, dZimg, Yimg, dXimg = h2i.HImage.from_tensor(X), h2i.HImage.from_tensor(dZ), h2i.HImage(M, 512, DEV), h2i.HImage(M, 512, DEV)
    wset = h2i.WeightSet()
    imgs = ops.WeightImages()
    total = 0.0
    for t in (X, dZ):                       # two-term fp16 path: the operands bring their amax (no fallback launch inside the timed loops)
        ops.amax_static(t)
    with imgs:
        ops.linear_fwd(X, W, b, Y, "relu", mask=mask, split=True)
        ops.linear_dgrad(dZ, W, dX, None, "relu", mask=mask, split=True)
        h2i.linear_fwd(Ximg, W, b, Y, None, "relu", mask=mask, wset=wset)
        h2i.linear_dgrad(dZimg, W, None, dXimg, mask=mask, wset=wset)
    print(f"idle: {power_now()}")
    with imgs:
        total += measure("forward 512 x 512, converting kernel (ReLU input: half zeros)", lambda: ops.linear_fwd(X, W, b, Y, "relu", mask=mask, split=True), 240)
        total += measure("data gradient 512 x 512, converting kernel", lambda: ops.linear_dgrad(dZ, W, dX, None, "relu", mask=mask, split=True), 240)
        measure("forward 512 x 512, operand images -> fp32", lambda: h2i.linear_fwd(Ximg, W, b, Y, None, "relu", mask=mask, wset=wset), 0)
        measure("forward 512 x 512, operand images -> image", lambda: h2i.linear_fwd(Ximg, W, b, None, Yimg, "relu", mask=mask, wset=wset), 240)
        measure("data gradient 512 x 512, operand images -> image", lambda: h2i.linear_dgrad(dZimg, W, None, dXimg, mask=mask, wset=wset), 240)
    s3, i3 = [], []
    for N, K in ((512, 512), (512, 512), (693, 512)):
        dz, x = rn(M, N) * 1e-3, torch.relu(rn(M, K))
        dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        ops.amax_static(dz)
        ops.amax_static(x)
        s3.append((dz, x, dW, db))
        i3.append((h2i.HImage.from_tensor(dz), h2i.HImage.from_tensor(x), dW, 0, db))
    ws = ops.workspace(ops.wgrad_group_workspace_bytes(s3, M, split=True), DEV)
    wi = ops.workspace(h2i.wgrad_group_workspace_bytes(i3, M), DEV)
    with imgs:
        total += measure("grouped weight gradient, 3 wide layers, converting kernel", lambda: ops.wgrad_group(s3, M, ws, split=True), 80)
    measure("grouped weight gradient, 3 wide layers, operand images", lambda: h2i.wgrad_group(i3, M, wi), 80)
    Xn, Wn, bn, Yn = rn(M, 265), rn(128, 265) / 16.0, rn(128), torch.empty(M, 128, device=DEV)
    total += measure("narrow forward 24576 x 128 x 265 (single-pass fp32 MFMA)", lambda: ops.linear_fwd(Xn, Wn, bn, Yn, "relu", split=False), 280)
    fs = torch.empty(M, 512, device=DEV)
    measure("forward 512 x 512, single-pass fp32 MFMA kernel", lambda: ops.linear_fwd(X, W, b, fs, "relu", split=False), 0)
    cp = torch.empty(M * 512, device=DEV)
    measure("device copy 50 MB (HBM read + write)", lambda: cp.copy_(Y.view(-1)), 0)
    print(f"sum of the rows with a per-step count (the wide GEMM classes scaled to the step's 240 + 240 + 80 + 280 launches): {total:.1f} J;"
          " one bench step at ~1380 W x 62 ms = 86 J")


if __name__ == "__main__":
    main()
