"""Small fixed workloads for rocprofv3 runs:  prof_target.py fwd|dgrad|wgrad|scorer|update"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import foothold, ops, synthetic as S  # noqa: E402

DEV = "cuda:0"
what = sys.argv[1] if len(sys.argv) > 1 else "fwd"
M, N, K = 24576, 512, 693
if what in ("fwd", "dgrad", "wgrad"):
    X = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    Y = torch.empty(M, N, device=DEV)
    dZ = torch.randn(M, N, device=DEV)
    dX = torch.empty(M, K, device=DEV)
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ws = torch.empty(ops.wgrad_workspace_bytes(M, N, K) // 4, device=DEV)
    for _ in range(10):
        if what == "fwd":
            ops.linear_fwd(X, W, b, Y, "relu")
        elif what == "dgrad":
            ops.linear_dgrad(dZ, W, dX, X, "relu")
        else:
            ops.linear_wgrad(dZ, X, dW, db, ws)
elif what == "scorer":
    inp = {k: v.to(DEV) for k, v in S.scorer_inputs(98304, seed=1).items()}
    for _ in range(10):
        foothold.plan(inp["measured_heights"], inp["root_states"], inp["thigh_pos"], inp["commands"])
torch.cuda.synchronize()
