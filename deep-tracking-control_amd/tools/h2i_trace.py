"""Per-workgroup timeline of one image-operand GEMM launch (dtc_h2i_trace):  h2i_trace.py [N] [K] [reps]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, h2i, ops  # noqa: E402

DEV = "cuda:0"
M = 24576
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K = int(sys.argv[2]) if len(sys.argv) > 2 else 512
X, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / K ** 0.5, torch.randn(N, device=DEV)
Xi, Yi, wset = h2i.HImage.from_tensor(X), h2i.HImage(M, N, DEV), h2i.WeightSet()
mask = ops.relu_mask(M, N, DEV) if N % 128 == 0 else None
grid = 8 * ((M // 128 + 7) // 8) * ((N + 127) // 128)
buf = torch.zeros(4 * grid, dtype=torch.int64, device=DEV)
run = lambda: h2i.linear_fwd(Xi, W, b, None, Yi, "relu" if mask is not None else None, mask=mask, wset=wset)
for _ in range(5):
    run()
torch.cuda.synchronize()
_ffi.lib().dtc_h2i_trace(buf.data_ptr())
for rep in range(1):
    buf.zero_()
    torch.cuda.synchronize()
    run()
    run()          # the SECOND of two back-to-back launches overwrites the first: steady-state launch
    torch.cuda.synchronize()
    t = buf.cpu().numpy().reshape(grid, 4)
    t = t[t[:, 0] > 0]
    t0 = t[:, 0].min()
    s, k, e = (t[:, 0] - t0) / 100.0, (t[:, 1] - t0) / 100.0, (t[:, 2] - t0) / 100.0
    hw = t[:, 3]
    xcd_like = (hw >> 0) & 0xffffffff
    print(f"rep {rep}: {len(t)} workgroups; kernel span {e.max():.1f} us; start: median {np.median(s):.1f}, p90 {np.percentile(s, 90):.1f}, max {s.max():.1f} us; "
          f"lifetime: median {np.median(e - s):.1f}, min {(e - s).min():.1f}, max {(e - s).max():.1f} us; K loop median {np.median(k - s):.1f}, epilogue median {np.median(e - k):.1f} us; "
          f"end: p10 {np.percentile(e, 10):.1f}, median {np.median(e):.1f}, p90 {np.percentile(e, 90):.1f} us")
    full = buf.cpu().numpy().reshape(grid, 4)
    xcd = np.arange(grid) % 8
    ok = full[:, 0] > 0
    for x in range(8):
        m = ok & (xcd == x)
        kk, ee = (full[m, 1] - full[m, 0]) / 100.0, (full[m, 2] - full[m, 1]) / 100.0
        print(f"   xcd {x}: K loop median {np.median(kk):.1f} (min {kk.min():.1f}, max {kk.max():.1f}), epilogue median {np.median(ee):.1f} (max {ee.max():.1f}), last end {((full[m, 2] - t0) / 100.0).max():.1f}")
    cu = (full[:, 3] >> 8) & 0xf
    se = (full[:, 3] >> 13) & 0x7
    m0 = ok & (xcd == 0)
    life = (full[:, 2] - full[:, 0]) / 100.0
    print("   xcd 0 lifetimes by (se, cu):", {(int(a), int(b)): [round(float(v), 1) for v in life[m0 & (se == a) & (cu == b)]] for a in np.unique(se[m0]) for b in np.unique(cu[m0 & (se == a)])})
    # K-loop time by column tile / by position of the row tile in its XCD's list (map_tile: block b -> xcd = b & 7, j = b >> 3, tc = j % col_tiles,
    # row tile = xcd + 8 * (j // col_tiles)): does the spread follow the tile (data placement, first reader of a row tile) or the CU?
    ct = (N + 127) // 128
    b_all = np.arange(grid)
    tc_all, loc_all = (b_all >> 3) % ct, (b_all >> 3) // ct
    kl = (full[:, 1] - full[:, 0]) / 100.0
    print("   K loop median by column tile:", [round(float(np.median(kl[ok & (tc_all == c)])), 1) for c in range(ct)])
    print("   K loop median by row-tile position in the XCD (0..):", [round(float(np.median(kl[ok & (loc_all == i)])), 1) for i in range(int(loc_all[ok].max()) + 1)])
    print("   K loop spread inside one row tile (max - min over its column tiles), median over row tiles:",
          round(float(np.median([np.ptp(kl[ok & (loc_all == i) & (xcd == x)]) for i in range(int(loc_all[ok].max()) + 1) for x in range(8) if (ok & (loc_all == i) & (xcd == x)).any()])), 1))
    hwid = full[:, 3]
    print("   distinct hw-id words:", len(np.unique(hwid[ok])), "examples:", [hex(int(v)) for v in np.unique(hwid[ok])[:6]])
    for word in np.unique(hwid[ok])[:0]:
        pass
    by_hw = {}
    place = ((hwid >> 32) & 0xf) << 16 | (hwid & 0xff00)          # (xcc, se, sh, cu)
    for v, kk in zip(place[ok], kl[ok]):
        by_hw.setdefault(int(v), []).append(float(kk))
    print("   distinct (xcc, se, sh, cu) places:", len(by_hw), "; workgroups per place:", sorted(set(len(v) for v in by_hw.values())),
          "; K-loop spread INSIDE a place (max - min), median:", round(float(np.median([max(v) - min(v) for v in by_hw.values()])), 1),
          "; spread of the place means:", round(float(min(np.mean(v) for v in by_hw.values())), 1), "...", round(float(max(np.mean(v) for v in by_hw.values())), 1))
    sp = sorted((np.mean(v), len(v), hex(k)) for k, v in by_hw.items())
    print("   K loop mean by hw-id word: fastest", [(round(a, 1), n, h) for a, n, h in sp[:5]], "slowest", [(round(a, 1), n, h) for a, n, h in sp[-5:]])
    # starts by order of blockIdx (dispatch order)
    order = np.argsort(t[:, 0])
    print("   start times of every 64th workgroup in dispatch order:", np.round(np.sort(s)[::64], 1).tolist())
_ffi.lib().dtc_h2i_trace(None)
