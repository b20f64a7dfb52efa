"""Fixed workloads on the image-operand kernels for rocprofv3 / timing runs:  h2i_probe.py fwd|dgrad|wgrad|all [zero] [time]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import h2i, ops  # noqa: E402

DEV = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
zero = "zero" in sys.argv
M, N, K = 24576, 512, 512
mk = (lambda *s: torch.zeros(*s, device=DEV)) if zero else (lambda *s: torch.randn(*s, device=DEV))
X, W, b = mk(M, K), mk(N, K) / K ** 0.5, mk(N)
Xi, Yi, wset = h2i.HImage.from_tensor(X), h2i.HImage(M, N, DEV), h2i.WeightSet()
dZi, dXi = h2i.HImage.from_tensor(mk(M, N)), h2i.HImage(M, K, DEV)
mask = ops.relu_mask(M, N, DEV)
dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
jobs = [(dZi, Xi, dW, 0, db)] * 3
wws = ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, M), DEV)
fns = dict(fwd=lambda: h2i.linear_fwd(Xi, W, b, None, Yi, "relu", mask=mask, wset=wset),
           fwd32=lambda: h2i.linear_fwd(Xi, W, b, X, None, "relu", wset=wset),
           dgrad=lambda: h2i.linear_dgrad(dZi, W, None, dXi, mask=mask, wset=wset),
           wgrad=lambda: h2i.wgrad_group(jobs, M, wws))
sel = list(fns) if what == "all" else [what]
for k in sel:
    for _ in range(3):
        fns[k]()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40 if "time" in sys.argv else 10
    e0.record()
    for _ in range(n):
        fns[k]()
    e1.record()
    torch.cuda.synchronize()
    print(f"{k}{' zero' if zero else ''}: {e0.elapsed_time(e1) / n * 1e3:.1f} us")
