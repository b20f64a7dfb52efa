"""Digest of the image-operand GEMM results on fixed operands (forward image + fp32, data gradient): run under two builds (DTC_LIB) to
check that a kernel variant is bit-identical."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import h2i, ops  # noqa: E402

DEV = "cuda:0"
g = torch.Generator(device=DEV).manual_seed(7)
for M, N, K in ((24576, 512, 512), (24576, 693, 512), (4096, 512, 752), (777, 256, 584)):
    X = torch.randn(M, K, device=DEV, generator=g) * torch.exp(3 * torch.randn(M, 1, device=DEV, generator=g))
    W, b = torch.randn(N, K, device=DEV, generator=g) / K ** 0.5, torch.randn(N, device=DEV, generator=g)
    Xi, Yi, wset = h2i.HImage.from_tensor(X), h2i.HImage(M, N, DEV), h2i.WeightSet()
    Y = torch.empty(M, N, device=DEV)
    mask = ops.relu_mask(M, N, DEV) if (N % 128 == 0 and M % 128 == 0) else None
    h2i.linear_fwd(Xi, W, b, Y, Yi, "relu" if mask is not None else None, mask=mask, wset=wset)
    dZi, dXi = h2i.HImage.from_tensor(torch.randn(M, N, device=DEV, generator=g)), h2i.HImage(M, K, DEV)
    dX = torch.empty(M, K, device=DEV)
    h2i.linear_dgrad(dZi, W, dX, dXi, mask=None, wset=wset)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for t in (Y, Yi.to_tensor(), dX, dXi.to_tensor()):
        h.update(t.cpu().numpy().tobytes())
    print(M, N, K, h.hexdigest()[:16])
