"""Per-workgroup, per-layer timeline of one chain launch (dtc_h2i_trace):  h2i_chain_trace.py [layers]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, h2i, ops  # noqa: E402

DEV = "cuda:0"
M, NL = 24576, int(sys.argv[1]) if len(sys.argv) > 1 else 3
Ws = [torch.randn(512, 512, device=DEV) / 23 for _ in range(NL)]
bs = [torch.randn(512, device=DEV) for _ in range(NL)]
imgs = [h2i.HImage.from_tensor(torch.randn(M, 512, device=DEV))] + [h2i.HImage(M, 512, DEV) for _ in range(NL)]
masks = [ops.relu_mask(M, 512, DEV) for _ in range(NL)]
wset, ch = h2i.WeightSet(), h2i.Chain(M, DEV)


def run():
    for i in range(NL):
        h2i.linear_fwd(imgs[i], Ws[i], bs[i], None, imgs[i + 1], "relu", mask=masks[i], wset=wset, chain=ch)
    ch.run()


for _ in range(4):
    run()
torch.cuda.synchronize()
grid = 768
buf = torch.zeros(4 * grid * NL, dtype=torch.int64, device=DEV)
_ffi.lib().dtc_h2i_trace(buf.data_ptr())
run()
run()
torch.cuda.synchronize()
_ffi.lib().dtc_h2i_trace(None)
t = buf.cpu().numpy().reshape(NL, grid, 4).astype(np.float64)
t0 = t[0, :, 3][t[0, :, 3] > 0].min()
for l in range(NL):
    arr, st, kd, en = ((t[l, :, c] - t0) / 100.0 for c in (3, 0, 1, 2))
    print(f"layer {l}: arrive median {np.median(arr):.1f} (p10 {np.percentile(arr, 10):.1f}, p90 {np.percentile(arr, 90):.1f}); wait median {np.median(st - arr):.1f} (p90 {np.percentile(st - arr, 90):.1f}, max {(st - arr).max():.1f}); "
          f"K loop median {np.median(kd - st):.1f} (p10 {np.percentile(kd - st, 10):.1f}, p90 {np.percentile(kd - st, 90):.1f}); epilogue median {np.median(en - kd):.1f}; end median {np.median(en):.1f}, max {en.max():.1f}")
