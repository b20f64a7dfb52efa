"""Per-kernel micro-benchmarks on one MI355X (HIP-event timing through dtc_prof_*).

    python deep-tracking-control_amd/tools/microbench.py [gemm] [scorer] [gae]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, foothold, ops, synthetic as S  # noqa: E402

DEV = "cuda:0"


def timed(fn, iters=50, warm=25):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def gemm():
    M = 24576
    shapes = [(512, 693), (512, 512), (693, 512), (512, 584), (512, 752), (256, 512), (128, 256), (64, 531), (128, 265),
              (12, 128), (35, 64)]
    for N, K in shapes:
        X = torch.randn(M, K, device=DEV)
        W = torch.randn(N, K, device=DEV) / K ** 0.5
        b = torch.randn(N, device=DEV)
        Y = torch.empty(M, N, device=DEV)
        dZ = torch.randn(M, N, device=DEV)
        dX = torch.empty(M, K, device=DEV)
        dW = torch.empty(N, K, device=DEV)
        db = torch.empty(N, device=DEV)
        ws = torch.empty(ops.wgrad_workspace_bytes(M, N, K) // 4, device=DEV)
        fl = 2.0 * M * N * K
        t0 = timed(lambda: torch.mm(X, W.t()))
        t1 = timed(lambda: ops.linear_fwd(X, W, b, Y, "relu"))
        t2 = timed(lambda: ops.linear_dgrad(dZ, W, dX, X, "relu"))
        t3 = timed(lambda: ops.linear_wgrad(dZ, X, dW, db, ws))
        t4 = timed(lambda: torch.mm(X, W.t()))
        print(f"M={M} N={N:4d} K={K:4d}  fwd {t1*1e3:8.1f} us {fl/t1/1e9:7.1f} TF | dgrad {t2*1e3:8.1f} us {fl/t2/1e9:7.1f} TF"
              f" | wgrad {t3*1e3:8.1f} us {fl/t3/1e9:7.1f} TF | torch.mm {t4*1e3:8.1f} us {fl/t4/1e9:7.1f} TF (first {fl/t0/1e9:6.1f})", flush=True)


def wgrad():
    M = 24576
    for N, K in [(512, 512), (512, 693), (512, 752), (512, 584), (256, 512), (693, 512)]:
        X = torch.randn(M, K, device=DEV)
        dZ = torch.randn(M, N, device=DEV)
        dW = torch.empty(N, K, device=DEV)
        db = torch.empty(N, device=DEV)
        ws = torch.empty(ops.wgrad_workspace_bytes(M, N, K) // 4, device=DEV)
        t = timed(lambda: ops.linear_wgrad(dZ, X, dW, db, ws))
        print(f"BN={os.environ.get('DTC_WGRAD_BN', '-')} BLOCKS={os.environ.get('DTC_WGRAD_BLOCKS', '-')} N={N} K={K}: "
              f"wgrad+reduce {t*1e3:7.1f} us {2.0*M*N*K/t/1e9:6.1f} TF", flush=True)


def ablate():
    """fwd GEMM with phases of the K loop removed (DTC_GEMM_ABLATE bits: 1 no global loads, 2 no MFMA,
    4 no LDS stores, 8 no barrier): which phase bounds the kernel?  Set per process via the env var."""
    M, N, K = 24576, 512, 693
    X = torch.randn(M, K, device=DEV)
    W = torch.randn(N, K, device=DEV) / K ** 0.5
    b = torch.randn(N, device=DEV)
    Y = torch.empty(M, N, device=DEV)
    t = timed(lambda: ops.linear_fwd(X, W, b, Y, "relu"), iters=30)
    print(f"ABLATE={os.environ.get('DTC_GEMM_ABLATE', '0')}: fwd {t*1e3:.1f} us  ({2.0*M*N*K/t/1e9:.1f} TF equivalent)", flush=True)


def scorer():
    for N in (4096, 32768, 98304):
        inp = {k: v.to(DEV) for k, v in S.scorer_inputs(N, seed=1).items()}
        t = timed(lambda: foothold.plan(inp["measured_heights"], inp["root_states"], inp["thigh_pos"], inp["commands"]))
        print(f"scorer N={N:6d}: {t*1e3:8.1f} us  {N*3096/t/1e9:8.1f} GB/s  {N/t*1e3:12.0f} env/s", flush=True)


def gae():
    for N in (4096, 32768):
        d = {k: v.to(DEV) for k, v in S.rollout(N, 24, seed=4).items() if k in ("rewards", "values", "dones", "last_values")}
        ret, adv = torch.empty(24, N, 1, device=DEV), torch.empty(24, N, 1, device=DEV)
        st = torch.zeros(4, dtype=torch.float64, device=DEV)

        def run():
            ops.gae(d["rewards"], d["values"], d["dones"], d["last_values"], 0.99, 0.95, ret, adv, st)
            ops.adv_sqdev(adv, st, 24 * N)
            ops.adv_normalize(adv, st, 24 * N)
        t = timed(run)
        print(f"gae+norm N={N}: {t*1e3:.1f} us  ({24*N*17/t/1e6:.1f} MB/s algorithmic)", flush=True)


if __name__ == "__main__":
    todo = sys.argv[1:] or ["gemm", "scorer", "gae"]
    print(torch.cuda.get_device_name(0))
    for t in todo:
        globals()[t]()
