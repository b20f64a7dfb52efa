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
elif what == "traffic":
    # HBM-traffic calibration + measurement for the GEMM loaders (dword buffer loads; MI355X_MICROARCH.md calibrates
    # FETCH_SIZE only for 16 B/lane reads).  Calibration case: ONE column tile (N = 64) and an X far larger than the
    # 256 MiB Infinity Cache, so every byte of X is fetched exactly once: known = M*K*4.
    Mc, Kc = 393216, 512
    Xc = torch.randn(Mc, Kc, device=DEV)
    Wc = torch.randn(64, Kc, device=DEV)
    Yc = torch.empty(Mc, 64, device=DEV)
    for _ in range(4):
        ops.linear_fwd(Xc, Wc, None, Yc, None)
    del Xc, Yc
    M, N, K = 24576, 512, 512
    X = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    Y = torch.empty(M, N, device=DEV)
    dZ = torch.randn(M, N, device=DEV)
    dX = torch.empty(M, K, device=DEV)
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ws = torch.empty(ops.wgrad_workspace_bytes(M, N, K) // 4, device=DEV)
    flush = torch.empty(96 * 1024 * 1024, device=DEV)            # 384 MB: evicts L2 / Infinity Cache between launches
    for _ in range(6):
        flush.fill_(1.0)
        ops.linear_fwd(X, W, b, Y, "relu")
        flush.fill_(2.0)
        ops.linear_dgrad(dZ, W, dX, X, "relu")
        flush.fill_(3.0)
        ops.linear_wgrad(dZ, X, dW, db, ws)
elif what == "scorer4096":
    # BASELINE configs[3]: one launch over 4096 envs x 4 legs (kernel-only duration by rocprofv3 --kernel-trace)
    inp = {k: v.to(DEV) for k, v in S.scorer_inputs(4096, seed=1).items()}
    for _ in range(30):
        foothold.plan(inp["measured_heights"], inp["root_states"], inp["thigh_pos"], inp["commands"])
elif what == "scorer":
    inp = {k: v.to(DEV) for k, v in S.scorer_inputs(98304, seed=1).items()}
    for _ in range(10):
        foothold.plan(inp["measured_heights"], inp["root_states"], inp["thigh_pos"], inp["commands"])
torch.cuda.synchronize()
