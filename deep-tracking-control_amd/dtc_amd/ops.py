"""Thin Python wrappers over the C ABI (one function per entry point of include/dtc_hip.h).

Every function takes device tensors, launches on torch's current stream and raises DtcError on
failure.  These are the building blocks used by storage/, modules/ and algorithms/.
"""
from __future__ import annotations

import torch

from . import _ffi
from ._ffi import ACT, check, cptr, lib, ptr, seg, segmat, stream

f32 = torch.float32


# ---------------------------------------------------------------- storage-side kernels
def gae(rewards, values, dones, last_values, gamma, lam, returns, advantages, stats):
    """[T,N,1] tensors; writes returns, un-normalised advantages, stats[0] = sum(adv)."""
    T, N = rewards.shape[0], rewards.shape[1]
    check(lib().dtc_gae(cptr(rewards, f32), cptr(values, f32), cptr(dones, torch.uint8), cptr(last_values, f32),
                        gamma, lam, cptr(returns, f32), cptr(advantages, f32), cptr(stats, torch.float64), T, N,
                        stream()), "dtc_gae")


def adv_sqdev(advantages, stats, count):
    check(lib().dtc_adv_sqdev(cptr(advantages, f32), cptr(stats, torch.float64), advantages.numel(), float(count),
                              stream()), "dtc_adv_sqdev")


def adv_normalize(advantages, stats, count):
    check(lib().dtc_adv_normalize(cptr(advantages, f32), cptr(stats, torch.float64), advantages.numel(),
                                  float(count), stream()), "dtc_adv_normalize")


def gather_rows(src: torch.Tensor, idx: torch.Tensor, out: torch.Tensor | None = None) -> torch.Tensor:
    """src[idx] for a contiguous 2-D+ tensor (rows = dim 0)."""
    assert src.is_contiguous() and idx.dtype == torch.int64 and idx.is_contiguous()
    rows = idx.numel()
    if out is None:
        out = torch.empty((rows,) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    row_bytes = src[0].numel() * src.element_size() if src.dim() > 1 else src.element_size()
    check(lib().dtc_gather_rows(ptr(src), ptr(idx), ptr(out), rows, row_bytes, stream()), "dtc_gather_rows")
    return out


# ---------------------------------------------------------------- dense layers
def as_segmat(x, idx=None):
    """Accept a plain 2-D tensor or an already built DtcSegMat."""
    if isinstance(x, _ffi.DtcSegMat):
        return x
    return segmat([seg(x, 0, x.shape[1])], idx)


def linear_fwd(X, W, b, Y, act=None, M=None):
    """Y = act(X W^T + b).  X: tensor or DtcSegMat; W [N,K]; Y [M,>=N] (row stride may exceed N)."""
    Xs = as_segmat(X)
    N, K = W.shape
    M = Y.shape[0] if M is None else M
    check(lib().dtc_linear_fwd(Xs, cptr(W, f32), cptr(b, f32) if b is not None else None, ptr(Y), Y.stride(0), M, N,
                               K, ACT[act], stream()), "dtc_linear_fwd")
    return Y


def linear_dgrad(dZ, W, dX, Xsaved=None, act=None, M=None):
    """dX = (dZ W) * act'(Xsaved); dX: tensor or DtcSegMat (destination)."""
    dXs = as_segmat(dX)
    N, K = W.shape
    M = dZ.shape[0] if M is None else M
    check(lib().dtc_linear_dgrad(ptr(dZ), dZ.stride(0), cptr(W, f32), dXs, ptr(Xsaved),
                                 Xsaved.stride(0) if Xsaved is not None else 0, M, N, K, ACT[act], stream()),
          "dtc_linear_dgrad")


def wgrad_workspace_bytes(M, N, K) -> int:
    return int(lib().dtc_linear_wgrad_workspace(M, N, K))


def linear_wgrad(dZ, X, dW, db, workspace, M=None):
    """dW = dZ^T X, db = colsum(dZ).  X: tensor or DtcSegMat."""
    Xs = as_segmat(X)
    N, K = dW.shape
    M = dZ.shape[0] if M is None else M
    need = wgrad_workspace_bytes(M, N, K)
    if workspace.numel() * workspace.element_size() < need:
        raise _ffi.DtcError(f"wgrad workspace too small: {workspace.numel() * workspace.element_size()} < {need}")
    check(lib().dtc_linear_wgrad(ptr(dZ), dZ.stride(0), Xs, cptr(dW, f32), cptr(db, f32) if db is not None else None,
                                 ptr(workspace), M, N, K, stream()), "dtc_linear_wgrad")
