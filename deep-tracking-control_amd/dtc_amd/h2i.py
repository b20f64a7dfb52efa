"""Block-scaled two-term fp16 operand images (include/dtc_hip.h "block-scaled ... operand images", csrc/h2i_core.hpp): the
representation of the wide layers' operands -- rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349 under
ppo.py:197-218, 252, 265, 289, 333.

    HImage            a [M, K] activation / gradient as (hi, lo) fp16 planes + one exponent per row and 128-column block
    WeightSet         the weight images of one trainer phase: learnt on the first pass, rebuilt by ONE launch per phase afterwards
    linear_fwd / linear_fwd_mse / linear_dgrad / wgrad_group: the products; results as fp32, as an image, or both

Nothing here keeps a scale on the host: every image carries its own exponents, computed by the kernel that wrote it.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _ffi
from ._ffi import ACT, check, cptr, lib, ptr, segmat, seg, stream

f32 = torch.float32


class HImage:
    """image(M, K): written by `pack` (from fp32) or by the epilogue of a product, read by LDS-DMA by every consumer."""

    def __init__(self, M, K, device):
        self.M, self.K = int(M), int(K)
        n = int(lib().dtc_h2i_bytes(self.M, self.K))
        if n <= 0 or n >= 1 << 31:
            raise _ffi.DtcError(f"operand image of a {M} x {K} matrix: {n} bytes (must be in (0, 2 GiB))")
        self.buf = torch.zeros((n + 7) // 8, dtype=torch.float64, device=device)      # (zeros: rows behind M read as nothing)
        # exponents start as "nothing here" (HI_EZERO): producers that write only their own rows (the latent / loss kernels) leave the
        # rows behind M of the last row tile marked empty, as the pack kernel and the GEMM epilogues do
        data = -(-self.M // 128) * -(-self.K // 16) * 8192
        self.buf.view(torch.int32)[data // 4: n // 4].fill_(0x7fff)

    @property
    def device(self):
        return self.buf.device

    def ptr(self):
        return self.buf.data_ptr()

    def pack(self, X, M=None):
        """self <- the fp32 operand X (tensor or DtcSegMat: segments side by side, row-gathered where asked)."""
        Xs = X if isinstance(X, _ffi.DtcSegMat) else segmat([seg(X, 0, X.shape[1])])
        assert Xs.cols == self.K
        check(lib().dtc_h2i_pack(Xs, self.M if M is None else M, self.ptr(), stream()), "dtc_h2i_pack")
        self._src = Xs                  # the launch reads the descriptor's tensors: keep them until the next pack
        return self

    @staticmethod
    def from_tensor(X):
        return HImage(X.shape[0], X.shape[1], X.device).pack(X)

    def to_tensor(self):
        """Decode (tests / debugging): fp32 [M, K]."""
        out = torch.empty(self.M, self.K, dtype=f32, device=self.device)
        check(lib().dtc_h2i_unpack(self.ptr(), self.M, self.K, ptr(out), out.stride(0), stream()), "dtc_h2i_unpack")
        return out

    def exps(self):
        """The exponent table [row tiles, k blocks, 128] (tests)."""
        rt, st = -(-self.M // 128), -(-self.K // 16)
        kb = -(-st // 8)
        off = rt * st * 8192
        return self.buf.view(torch.int32)[off // 4: off // 4 + rt * kb * 128].view(rt, kb, 128)


# ---------------------------------------------------------------- in-situ accuracy capture (bench.py: gemm_accuracy_in_situ)
CAPTURE = None       # dict while a capture runs: every product below is checked against fp64 on its ACTUAL operands, right after its launch


def capture_begin(per_key=2):
    """From here on every image-operand product records, for the first `per_key` calls of each (kind, shape): the largest element error
    and the largest per-row relative error against an fp64 product of the SAME operands (the decoded images), next to the single-pass
    fp32 MFMA kernel on those operands.  Synchronous and slow: run one serialised step inside it (overlap off)."""
    global CAPTURE
    CAPTURE = dict(per_key=per_key, rows={}, count={})


def capture_end():
    global CAPTURE
    out, CAPTURE = CAPTURE, None
    return out["rows"] if out else {}


def _errs(y, ref):
    y, ref = y.double(), ref.double()
    d = (y - ref).abs()
    den = ref.abs().amax(dim=1)
    live = den > 0
    row = float((d.amax(dim=1)[live] / den[live]).max()) if bool(live.any()) else 0.0
    return dict(max_rel=float(d.max() / ref.abs().max().clamp_min(1e-300)), row_rel=row)


def _cap_want(kind, shape):
    c = CAPTURE
    if c is None or c.get("busy"):
        return None
    key = f"{kind}[{'x'.join(str(v) for v in shape)}]"
    n = c["count"].get(key, 0)
    if n >= c["per_key"]:
        return None
    c["count"][key] = n + 1
    return key


def _cap_put(key, got, ref, native):
    rm = ref.abs().amax(dim=1)
    rm = rm[rm > 0]
    e = dict(h2i=_errs(got, ref), fp32_mfma=_errs(native, ref) if native is not None else None,
             ref_rows_span=float(rm.log10().max() - rm.log10().min()) if rm.numel() else 0.0, zero_rows=int(ref.shape[0] - rm.numel()))
    CAPTURE["rows"].setdefault(key, []).append(e)


def unpack_sign_record(mask, M, N):
    """The ReLU sign record of an [M, N] layer output (dtc_linear_fwd_mask layout) as a bool matrix."""
    w = mask.view(torch.int16)[:(M // 32) * 2 * N].view(M // 32, 2, N).to(torch.int32) & 0xffff
    out = torch.zeros(M // 32, 32, N, dtype=torch.bool, device=mask.device)
    for r in range(16):
        for half in range(2):
            out[:, (r & 3) + 8 * (r >> 2) + 4 * half, :] = ((w[:, half, :] >> r) & 1).bool()
    return out.view(M, N)


def _act64(v, act):
    if act in ("relu", "crelu"):
        return torch.relu(v)
    if act == "elu":
        return torch.nn.functional.elu(v)
    assert act in (None, "none"), act
    return v


def _operand(X):
    """HImage | list of HImages (side by side along the reduction) -> DtcH2iOperand"""
    imgs = [X] if isinstance(X, HImage) else list(X)
    op = _ffi.DtcH2iOperand()
    op.nseg = len(imgs)
    for i, im in enumerate(imgs):
        op.width[i], op.img[i] = im.K, im.ptr()
    op._keep = imgs
    return op, imgs


def _wjob(job, W, trans, rows, ranges, img_ptr):
    job.W, job.ld, job.img = cptr(W, f32), W.stride(0), img_ptr
    job.trans, job.nrows, job.nseg = int(trans), len(rows), len(ranges)
    for i, (r0, nr) in enumerate(rows):
        job.r0[i], job.nr[i] = int(r0), int(nr)
    for i, (c0, cw) in enumerate(ranges):
        job.c0[i], job.cw[i] = int(c0), int(cw)


class WeightSet:
    """Weight images of ONE trainer phase.  `get(W, trans, rows, ranges)` (rows: up to two (first row, count) ranges) returns the image of that product's weight operand;
    the first request builds it on the spot (and remembers the job), `rebuild()` -- called once per phase, after the optimiser wrote
    the weights and before the lanes fork -- builds all remembered images with one grouped launch."""

    def __init__(self):
        self.entries = {}           # key -> (image buffer, job fields, W)
        self.jobs = None

    def get(self, W, trans, rows, ranges):
        rows = [(int(a), int(b)) for a, b in rows]
        key = (W.data_ptr(), W.stride(0), int(trans), tuple(rows), tuple((int(a), int(b)) for a, b in ranges))
        e = self.entries.get(key)
        if e is None:
            job = (_ffi.DtcH2iWJob * 1)()
            _wjob(job[0], W, trans, rows, ranges, None)
            n = int(lib().dtc_h2i_wimage_bytes(job))
            if n <= 0:
                raise _ffi.DtcError("dtc_h2i_wimage_bytes: bad weight-image job")
            buf = torch.zeros((n + 7) // 8, dtype=torch.float64, device=W.device)
            job[0].img = buf.data_ptr()
            check(lib().dtc_h2i_wimage_group(job, 1, stream()), "dtc_h2i_wimage_group")
            e = self.entries[key] = (buf, (W, trans, rows, ranges))
            self.jobs = None
        return e[0]

    def rebuild(self):
        if not self.entries:
            return
        if self.jobs is None:
            self.jobs = (_ffi.DtcH2iWJob * len(self.entries))()
            for j, (buf, (W, trans, rows, ranges)) in zip(self.jobs, self.entries.values()):
                _wjob(j, W, trans, rows, ranges, buf.data_ptr())
        check(lib().dtc_h2i_wimage_group(self.jobs, len(self.jobs), stream()), "dtc_h2i_wimage_group")


_ONE_SHOT = WeightSet()      # calls outside a trainer phase (tests, direct use): images built per call


def _fwd_wimage(W, imgs, col_ranges, wset):
    """The forward weight image for row operand `imgs`: col_ranges[i] = first column of W that meets image i (default: side by side)."""
    if col_ranges is None:
        col_ranges, c = [], 0
        for im in imgs:
            col_ranges.append(c)
            c += im.K
    ranges = [(c0, im.K) for c0, im in zip(col_ranges, imgs)]
    ws = wset if wset is not None else WeightSet()
    return ws.get(W, 0, [(0, W.shape[0])], ranges), ws


def linear_fwd(X, W, b, Y=None, Yimg=None, act=None, mask=None, wset=None, cols=None):
    """Y = act(X W^T + b).  X: HImage or list of HImages; results: fp32 `Y` [M, >= N] and / or the HImage `Yimg` [M, N].
    `cols`: first column of W for each image of X (default: the images side by side from column 0)."""
    op, imgs = _operand(X)
    N = W.shape[0]
    M = imgs[0].M
    assert (Y is not None or Yimg is not None) and (Yimg is None or (Yimg.M, Yimg.K) == (M, N))
    wimg, ws = _fwd_wimage(W, imgs, cols, wset)
    check(lib().dtc_linear_fwd_h2i(op, ptr(wimg), cptr(b, f32) if b is not None else None, ptr(Y) if Y is not None else None,
                                   Y.stride(0) if Y is not None else 0, Yimg.ptr() if Yimg is not None else None,
                                   ptr(mask) if mask is not None else None, M, N, ACT[act], stream()), "dtc_linear_fwd_h2i")
    key = _cap_want("fwd", (M, N, sum(im.K for im in imgs)))
    if key:
        from . import ops
        CAPTURE["busy"] = True
        starts = cols if cols is not None else [sum(im.K for im in imgs[:i]) for i in range(len(imgs))]
        Xf = torch.zeros(M, W.shape[1], dtype=f32, device=W.device)
        for c0, im in zip(starts, imgs):
            Xf[:, c0:c0 + im.K] = im.to_tensor()
        ref = _act64(Xf.double() @ W.double().T + (b.double() if b is not None else 0.0), act)
        nat = torch.empty(M, N, dtype=f32, device=W.device)
        ops.linear_fwd(Xf, W, b, nat, act, split=False)
        _cap_put(key, Y[:, :N] if Y is not None else Yimg.to_tensor(), ref, nat)
        CAPTURE["busy"] = False
    return ws


def linear_fwd_chain(layers, wset=None):
    """Up to three forward layers of at most 512 output columns each (round 6: several column tiles per layer, run one after the other by the
    row tile's workgroup) in ONE launch (dtc_linear_fwd_chain_h2i): `layers` = list of dicts
    with the keyword arguments of linear_fwd (X, W, b, Y, Yimg, act, mask, cols); layer i + 1's X must be layer i's Yimg.  Bit for bit
    the per-layer calls.  (A capture in flight takes the per-layer calls: it checks every product on its own.)"""
    if CAPTURE is not None or len(layers) == 1:
        ws = wset
        for L in layers:
            ws = linear_fwd(L["X"], L["W"], L.get("b"), L.get("Y"), L.get("Yimg"), L.get("act"), L.get("mask"), ws if ws is not None else wset, L.get("cols"))
        return ws
    arr = (_ffi.DtcH2iFwdLayer * len(layers))()
    keep, ws = [], wset
    M = None
    for a, L in zip(arr, layers):
        op, imgs = _operand(L["X"])
        M = imgs[0].M if M is None else M
        W, Y, Yimg, mask, b = L["W"], L.get("Y"), L.get("Yimg"), L.get("mask"), L.get("b")
        wimg, ws = _fwd_wimage(W, imgs, L.get("cols"), ws if ws is not None else wset)
        a.X, a.wimg, a.b = op, ptr(wimg), (cptr(b, f32) if b is not None else None)
        a.Y, a.ldy = (ptr(Y) if Y is not None else None), (Y.stride(0) if Y is not None else 0)
        a.Yimg, a.relu_mask = (Yimg.ptr() if Yimg is not None else None), (ptr(mask) if mask is not None else None)
        a.N, a.act = W.shape[0], ACT[L.get("act")]
        keep.append((op, imgs, wimg))
    check(lib().dtc_linear_fwd_chain_h2i(arr, len(layers), M, stream()), "dtc_linear_fwd_chain_h2i")
    return ws


def linear_dgrad_chain(layers, wset=None):
    """Up to three data-gradient layers whose windows are at most 512 columns wide in ONE launch (dtc_linear_dgrad_chain_h2i): `layers` =
    list of dicts with the arguments of linear_dgrad (dZimg, W, dX, dXimg, window (one range), add, Xsaved, act, mask); layer i + 1's
    dZimg must be layer i's dXimg.  Bit for bit the per-layer calls."""
    if CAPTURE is not None or len(layers) == 1:
        ws = wset
        for L in layers:
            ws = linear_dgrad(L["dZimg"], L["W"], L.get("dX"), L.get("dXimg"), L.get("window"), L.get("add"), L.get("Xsaved"), L.get("act"),
                              L.get("mask"), ws if ws is not None else wset)
        return ws
    arr = (_ffi.DtcH2iDgradLayer * len(layers))()
    keep = []
    ws = wset if wset is not None else WeightSet()
    M = layers[0]["dZimg"].M
    for a, L in zip(arr, layers):
        dZimg, W, dX, dXimg, mask, add, Xs, act = (L["dZimg"], L["W"], L.get("dX"), L.get("dXimg"), L.get("mask"), L.get("add"), L.get("Xsaved"),
                                                   L.get("act"))
        N, K = W.shape
        win = L.get("window") or (0, K)
        assert isinstance(win[0], int) and dZimg.K == N
        wimg = ws.get(W, 1, [win], [(0, N)])
        dXs = None
        if dX is not None:
            dXs = dX if isinstance(dX, _ffi.DtcSegMat) else segmat([seg(dX, 0, win[1])])
        a.dZimg, a.wimgT, a.dX = dZimg.ptr(), ptr(wimg), (C.pointer(dXs) if dXs is not None else None)
        a.dXimg, a.img_cols = (dXimg.ptr() if dXimg is not None else None), (dXimg.K if dXimg is not None else 0)
        a.add, a.ld_add = (ptr(add) if add is not None else None), (add.stride(0) if add is not None else 0)
        a.Xsaved, a.ldxs = (ptr(Xs) if (mask is None and Xs is not None) else None), (Xs.stride(0) if Xs is not None else 0)
        a.relu_mask, a.N, a.Kwin = (ptr(mask) if mask is not None else None), N, win[1]
        a.act = ACT[act] if mask is None else ACT["relu"]
        keep.append((dXs, wimg))
    check(lib().dtc_linear_dgrad_chain_h2i(arr, len(layers), M, stream()), "dtc_linear_dgrad_chain_h2i")
    return ws


def mse_parts(M, N) -> int:
    return int(lib().dtc_linear_fwd_mse_h2i_parts(M, N))


def linear_fwd_mse(X, W, b, target, tcol0, tidx, dY, dYimg, part, wset=None):
    """The output layer fused with its MSE against target[tidx, tcol0:tcol0 + N]; dL/dY as fp32 `dY` and / or HImage `dYimg`;
    returns the number of partial sums written to `part` (float64)."""
    op, imgs = _operand(X)
    N = W.shape[0]
    M = imgs[0].M
    n = mse_parts(M, N)
    assert part.numel() >= n and part.dtype == torch.float64
    wimg, ws = _fwd_wimage(W, imgs, None, wset)
    check(lib().dtc_linear_fwd_mse_h2i(op, ptr(wimg), cptr(b, f32) if b is not None else None, cptr(target, f32), target.stride(0),
                                       target.shape[0], tcol0, cptr(tidx, torch.int64), 2.0 / (M * N), ptr(dY) if dY is not None else None,
                                       dY.stride(0) if dY is not None else 0, dYimg.ptr() if dYimg is not None else None, ptr(part), M, N,
                                       stream()), "dtc_linear_fwd_mse_h2i")
    key = _cap_want("fwd_mse", (M, N, sum(im.K for im in imgs)))
    if key:
        CAPTURE["busy"] = True
        Xf = torch.cat([im.to_tensor() for im in imgs], dim=1)
        ref = ((Xf.double() @ W.double().T + (b.double() if b is not None else 0.0)) - target[tidx][:, tcol0:tcol0 + N].double()) * (2.0 / (M * N))
        _cap_put(key, dY[:, :N] if dY is not None else dYimg.to_tensor(), ref, None)
        CAPTURE["busy"] = False
    return n


def linear_dgrad(dZimg, W, dX=None, dXimg=None, window=None, add=None, Xsaved=None, act=None, mask=None, wset=None):
    """dX[:, window] = ((dZ W[:, window]) + add) * act'(.).  dZimg: HImage [M, N]; window = (first column, width) of W's columns, or a
    list of up to two such ranges computed one after the other (all but the last a multiple of 128 wide; default: all columns); results
    over the window: fp32 destination `dX` (tensor or DtcSegMat covering the window) and / or the HImage `dXimg` of the window's first
    dXimg.K columns; `add`: fp32 [M, >= width] added to the product before anything is stored."""
    N, K = W.shape
    wins = [(0, K)] if window is None else ([window] if isinstance(window[0], int) else list(window))
    kw = sum(w for _, w in wins)
    M = dZimg.M
    assert dZimg.K == N and (dX is not None or dXimg is not None) and (dXimg is None or (dXimg.M == M and dXimg.K <= kw))
    ws = wset if wset is not None else WeightSet()
    wimg = ws.get(W, 1, wins, [(0, N)])
    dXs = None
    if dX is not None:
        dXs = dX if isinstance(dX, _ffi.DtcSegMat) else segmat([seg(dX, 0, kw)])
    check(lib().dtc_linear_dgrad_h2i(dZimg.ptr(), N, ptr(wimg), kw, dXs, dXimg.ptr() if dXimg is not None else None,
                                     dXimg.K if dXimg is not None else 0,
                                     ptr(add) if add is not None else None, add.stride(0) if add is not None else 0,
                                     ptr(Xsaved) if (mask is None and Xsaved is not None) else None, Xsaved.stride(0) if Xsaved is not None else 0,
                                     ptr(mask) if mask is not None else None, M, ACT[act] if mask is None else ACT["relu"], stream()),
          "dtc_linear_dgrad_h2i")
    key = _cap_want("dgrad", (M, N, kw)) if (dXimg is not None or isinstance(dX, torch.Tensor)) else None
    if key:
        from . import ops
        CAPTURE["busy"] = True
        kc = dXimg.K if dXimg is not None else kw
        Wwin = torch.cat([W[:, c0:c0 + w] for c0, w in wins], dim=1)[:, :kc].contiguous()
        dZf = dZimg.to_tensor()
        ref = dZf.double() @ Wwin.double()
        if add is not None:
            ref = ref + add[:, :kc].double()
        if mask is not None:
            ref = ref * unpack_sign_record(mask, M, kw)[:, :kc]
        elif act not in (None, "none"):
            assert act == "elu"
            ys = Xsaved[:, :kc].double()
            ref = torch.where(ys > 0, ref, ref * (ys + 1.0))
        # the single-pass fp32 MFMA kernel on the same operands (added term and sign record applied as the image kernel's epilogue does)
        nat = torch.empty(M, kc, dtype=f32, device=W.device)
        if mask is None and add is None:
            ops.linear_dgrad(dZf, Wwin, nat, Xsaved[:, :kc] if act not in (None, "none") else None, act, split=False)
        else:
            ops.linear_dgrad(dZf, Wwin, nat, None, None, split=False)
            if add is not None:
                nat = add[:, :kc] + nat
            if mask is not None:
                nat = nat * unpack_sign_record(mask, M, kw)[:, :kc]
            elif act not in (None, "none"):
                ys32 = Xsaved[:, :kc]
                nat = torch.where(ys32 > 0, nat, nat * (ys32 + 1.0))
        _cap_put(key, dXimg.to_tensor() if dXimg is not None else dX[:, :kc], ref, nat)
        CAPTURE["busy"] = False
    return ws


def _wgrad_jobs(jobs):
    """jobs: list of (dZimg [M,N], Ximg [M,K], dW (the [N, ldw] gradient tensor), wcol0, db or None)"""
    arr = (_ffi.DtcWgradH2iJob * len(jobs))()
    for a, (dZimg, Ximg, dW, wcol0, db) in zip(arr, jobs):
        assert dZimg.M == Ximg.M and dW.shape[0] == dZimg.K and dW.stride(1) == 1 and wcol0 + Ximg.K <= dW.shape[1] + 15
        a.dZimg, a.Ximg, a.dW, a.db = dZimg.ptr(), Ximg.ptr(), ptr(dW), (cptr(db, f32) if db is not None else None)
        a.ldw, a.N, a.K, a.wcol0 = dW.stride(0), dZimg.K, min(Ximg.K, dW.shape[1] - wcol0), wcol0
    return arr


def wgrad_group_workspace_bytes(jobs, M) -> int:
    n = int(lib().dtc_wgrad_group_h2i_workspace(_wgrad_jobs(jobs), len(jobs), M))
    if n < 0:
        raise _ffi.DtcError(f"dtc_wgrad_group_h2i_workspace failed: {lib().dtc_last_error().decode()}")
    return n


def wgrad_group(jobs, M, workspace, stream_ptr=None):
    """dW[:, wcol0 : wcol0 + K] = dZ^T X, db = colsum(dZ) for up to 12 (dZimg, Ximg) pairs in one launch pair.
    Returns the objects the launch reads (keep them alive until it has run)."""
    arr = _wgrad_jobs(jobs)
    check(lib().dtc_wgrad_group_h2i(arr, len(jobs), M, ptr(workspace), stream() if stream_ptr is None else stream_ptr), "dtc_wgrad_group_h2i")
    if CAPTURE is not None and stream_ptr is None:
        from . import ops
        for dZimg, Ximg, dW, wcol0, db in jobs:
            K = min(Ximg.K, dW.shape[1] - wcol0)
            key = _cap_want("wgrad", (M, dZimg.K, K))
            if key:
                CAPTURE["busy"] = True
                dZf, Xf = dZimg.to_tensor(), Ximg.to_tensor()[:, :K].contiguous()
                ref = dZf.double().T @ Xf.double()
                natW, natb = torch.empty(dZimg.K, K, dtype=f32, device=dW.device), torch.empty(dZimg.K, dtype=f32, device=dW.device)
                nj = [(dZf, Xf, natW, natb)]
                ops.wgrad_group(nj, M, ops.workspace(ops.wgrad_group_workspace_bytes(nj, M, False), dW.device), split=False)
                _cap_put(key, dW[:, wcol0:wcol0 + K], ref, natW)
                CAPTURE["busy"] = False
    return jobs
