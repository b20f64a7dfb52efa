"""Narrow layers (< 128 columns) of the update on the operand-image kernels against the single-pass fp32 kernels, back to back at
M = 24576:  forward / data gradient per shape, small pack launches, and what the narrow weight gradients cost as extra jobs of a
wide image-operand group against their own fp32 grouped launch.  Timing only (random operands)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import h2i, ops  # noqa: E402

DEV = "cuda:0"
M = 24576


def timed(fn, n=40):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


wset = h2i.WeightSet()
rn = lambda *s: torch.randn(*s, device=DEV)       # noqa: E731
print("shape (N x K)      fwd h2i   fwd fp32   dgrad h2i   dgrad fp32   [us]")
for N, K in [(128, 265), (64, 128), (35, 64), (64, 531), (128, 64), (53, 128)]:
    X, W, b = rn(M, K), rn(N, K) / K ** 0.5, rn(N)
    Xi, Yi = h2i.HImage.from_tensor(X), h2i.HImage(M, N, DEV)
    Y = torch.empty(M, N, device=DEV)
    dZ = rn(M, N)
    dZi, dXi, dX = h2i.HImage.from_tensor(dZ), h2i.HImage(M, K, DEV), torch.empty(M, K, device=DEV)
    t = [timed(lambda: h2i.linear_fwd(Xi, W, b, None, Yi, "relu", wset=wset)),
         timed(lambda: ops.linear_fwd(X, W, b, Y, "relu", split=False)),
         timed(lambda: h2i.linear_dgrad(dZi, W, None, dXi, wset=wset)),
         timed(lambda: ops.linear_dgrad(dZ, W, dX, None, None, split=False))]
    print(f"{N:4d} x {K:4d}       " + "   ".join(f"{v:8.1f}" for v in t))

for w in (19, 35, 53, 265):
    X = rn(M, w)
    im = h2i.HImage(M, w, DEV)
    print(f"pack {w:4d} columns: {timed(lambda: im.pack(X)):.1f} us")

# weight gradients: the VAE step's decoder bucket (3 wide layers) with / without the CE-net decoder's narrow layers as extra jobs
def wjobs(shapes):
    out = []
    for N, K in shapes:
        out.append((h2i.HImage.from_tensor(rn(M, N)), h2i.HImage.from_tensor(rn(M, K)), torch.empty(N, K, device=DEV), 0,
                    torch.empty(N, device=DEV)))
    return out


wide = wjobs([(693, 512), (512, 512), (512, 512)])
narrow_dec = wjobs([(53, 128), (128, 64), (64, 531)])
narrow_enc = wjobs([(35, 64), (64, 128), (128, 265)])
wide_enc = wjobs([(512, 512), (512, 512), (512, 693)])
for name, jobs in (("decoder bucket, wide only", wide), ("decoder bucket + CE-net decoder", wide + narrow_dec),
                   ("encoder bucket, wide only", wide_enc), ("encoder bucket + CE-net encoder", wide_enc + narrow_enc)):
    ws = ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, M), DEV)
    print(f"{name:34s} {timed(lambda: h2i.wgrad_group(jobs, M, ws)):.1f} us")
for name, shapes in (("CE-net decoder, fp32 group", [(53, 128), (128, 64), (64, 531)]), ("CE-net encoder, fp32 group", [(35, 64), (64, 128), (128, 265)])):
    jobs = [(rn(M, N), rn(M, K), torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)) for N, K in shapes]
    ws = ops.workspace(ops.wgrad_group_workspace_bytes(jobs, M, False), DEV)
    print(f"{name:34s} {timed(lambda: ops.wgrad_group(jobs, M, ws, split=False)):.1f} us")
