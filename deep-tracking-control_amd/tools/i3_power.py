"""Round-4 probe: is the image-operand GEMM bound by the clock the chip sustains (DVFS) rather than by its instruction stream?  The same
kernel, same shapes, on all-zero operands against random ones (zeros toggle no multiplier bits: MI355X_MICROARCH.md 'DVFS give-back')."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import ops  # noqa: E402

DEV = "cuda:0"
M, N, K = 24576, 512, 512


def run(tag, X, W):
    b = torch.zeros(N, device=DEV)
    Y = torch.empty(M, N, device=DEV)
    Ximg = ops.AImage.from_tensor(X)
    imgs = ops.WeightImages()
    best = {}
    for rnd in range(3):
        with imgs:
            for name, fn in (("i3", lambda: ops.linear_fwd_img(Ximg, W, b, Y, None, "relu")), ("s3", lambda: ops.linear_fwd(X, W, b, Y, "relu", split=True)),
                             ("fp32 mfma", lambda: ops.linear_fwd(X, W, b, Y, "relu", split=False))):
                for _ in range(5):
                    fn()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(30):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                best[name] = min(best.get(name, 1e9), e0.elapsed_time(e1) * 1e3 / 30)
    print(f"{tag:28s} " + "   ".join(f"{k} {v:6.1f} us" for k, v in best.items()))


g = torch.Generator(device=DEV).manual_seed(1)
Xr, Wr = torch.randn(M, K, device=DEV, generator=g), torch.randn(N, K, device=DEV, generator=g) / 22.0
Xz, Wz = torch.zeros(M, K, device=DEV), torch.zeros(N, K, device=DEV)
Xb = Xr.bfloat16().float()          # values with 8 significant bits: planes 2 and 3 are zero
Wb = Wr.bfloat16().float()
only = sys.argv[1] if len(sys.argv) > 1 else None      # "random" / "zero": one case only (counter passes: tools/analysis/pmc_any.py)
if only == "random":
    run("random X, random W", Xr, Wr)
elif only == "zero":
    run("zero X, zero W", Xz, Wz)
else:
    run("random X, random W", Xr, Wr)
    run("zero X, zero W", Xz, Wz)
    run("random X, zero W", Xr, Wz)
    run("bf16-exact X and W", Xb, Wb)
    run("random X, random W (again)", Xr, Wr)
