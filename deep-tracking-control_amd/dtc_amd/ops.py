"""Thin Python wrappers over the C ABI (one function per entry point of include/dtc_hip.h).

Every function takes device tensors, launches on torch's current stream and raises DtcError on
failure.  These are the building blocks used by storage/, modules/ and algorithms/.
"""
from __future__ import annotations

import ctypes as C

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


def store_transition(copies, rewards=None, values=None, time_outs=None, gamma=0.0, rewards_dst=None):
    """RolloutStorage.add_transitions as one launch.  copies: list of (src [N, ...], dst [N, ...]) tensor pairs
    (src may be a broadcast view with row stride 0); rewards_dst = rewards + gamma * values * time_outs."""
    n_rows = copies[0][1].shape[0] if copies else rewards_dst.shape[0]
    items = (_ffi.DtcRowCopy * max(1, len(copies)))()
    for i, (src, dst) in enumerate(copies):
        assert dst.is_contiguous() and src.shape[0] == n_rows and dst.shape[0] == n_rows
        width = dst[0].numel() * dst.element_size()
        assert src.dtype == dst.dtype and src[0].numel() == dst[0].numel() and (src.dim() == 1 or src[0].is_contiguous())
        items[i].src, items[i].dst = src.data_ptr(), dst.data_ptr()
        items[i].src_stride_bytes = src.stride(0) * src.element_size()
        items[i].width_bytes = width
    check(lib().dtc_store_transition(items, len(copies), ptr(rewards), ptr(values), ptr(time_outs), float(gamma),
                                     ptr(rewards_dst), n_rows, stream()), "dtc_store_transition")


def store_transition_items(items, count, rewards, values, time_outs, gamma, rewards_dst, n_rows):
    """dtc_store_transition on an already marshalled DtcRowCopy array (RolloutStorage keeps one and re-points it)."""
    check(lib().dtc_store_transition(items, count, ptr(rewards), ptr(values), ptr(time_outs), float(gamma),
                                     ptr(rewards_dst), n_rows, stream()), "dtc_store_transition")


def history_roll(obs_history, obs, out, history_len, reset=None):
    """out <- cat(obs_history[:, D:], obs); `out` may be `obs_history` itself."""
    N, D = obs.shape
    check(lib().dtc_history_roll(cptr(obs_history, f32), cptr(obs, f32), cptr(out, f32),
                                 cptr(reset, torch.uint8) if reset is not None else None, N, history_len, D, stream()),
          "dtc_history_roll")
    return out


# ---------------------------------------------------------------- dense layers
def as_segmat(x, idx=None):
    """Accept a plain 2-D tensor or an already built DtcSegMat."""
    if isinstance(x, _ffi.DtcSegMat):
        return x
    return segmat([seg(x, 0, x.shape[1])], idx)


# split-precision GEMMs (csrc/gemm_s3.hip, csrc/wgrad_s3.hip): the wide layers of linear_fwd / linear_dgrad / linear_fwd_mse and every
# grouped weight gradient run on the bf16 matrix pipe with 3-term operand splits (fp32-level accuracy, tests/test_hip_split.py).
# DTC_GEMM_SPLIT=0 selects the single-pass fp32 MFMA kernels everywhere; `set_split()` switches at run time (tests, A/B runs)
import os as _os
SPLIT = _os.environ.get("DTC_GEMM_SPLIT", "2") != "0"
# representation of the split operands: two fp16 terms and three MFMA passes per product (DTC_GEMM_SPLIT=2, the default since round 4:
# include/dtc_hip.h "two-term fp16 path"; operands bring the amax of their tensor, class Amax below) or three bf16 terms and six passes
# (DTC_GEMM_SPLIT=1: no amax needed, twice the matrix-pipe work).  The recurrent kernels and the activation-image chain are bf16 x 3 only.
H2 = _os.environ.get("DTC_GEMM_SPLIT", "2") == "2"
AMAX_CHECK = _os.environ.get("DTC_AMAX_CHECK", "0") == "1"   # debug: re-derive every published amax at its use (synchronises)
_NOPUB_SMALL = False    # A/B aid: the latent / loss kernels publish no amax (their consumers compute it)
AMAX_STATS = {} if _os.environ.get("DTC_AMAX_STATS", "0") == "1" else None   # debug: (kind, shape) -> count of dtc_amax fallbacks
# routing thresholds (output columns / reduction length); swept with bench.py in round 3: 256 / 384 -> 70.2 ms, 128 / 128 -> 69.5 ms per
# step (the 128-column layers run longer per launch on 192 tiles of 128 x 128 -- 35 vs 28 us -- but on the second lane, under the wide GEMMs)
SPLIT_MIN_COLS = 128
SPLIT_MIN_RED = 128
# recurrent trainers: weight gradients over the valid slots of the padded trajectory layout only (0: over all T x R rows)
WGRAD_ROWS = True
WIMG_CHECK = _os.environ.get("DTC_WIMG_CHECK", "0") == "1"  # debug: re-derive every cached weight image at its use and compare
_NOT_NULL = 16                                             # stand-in address of a non-NULL operand block in a cached descriptor
WIMG = _os.environ.get("DTC_S3_WIMG", "1") != "0"          # the library's weight-image switch (csrc/gemm_s3.hip reads the same variable)
H2 = H2 and WIMG                                           # the fp16 kernels read their weights as images only


def set_split(on: bool, h2: bool | None = None):
    """Switch the split path on / off at run time; `h2` (when given) selects the operand representation (True: two fp16 terms)."""
    global SPLIT, H2
    SPLIT = bool(on)
    if h2 is not None:
        H2 = bool(h2) and WIMG            # (the fp16 kernels read their weights as images only: DTC_S3_WIMG=0 keeps the bf16 x 3 kernels)
    lib().dtc_set_gemm_split(int(SPLIT))


# ---------------------------------------------------------------- amax slots of the two-term fp16 path
class Amax:
    """Where the fp16 path's operands get the amax of their tensor from (include/dtc_hip.h: DtcSeg.amax).

    * published: a kernel that writes a WHOLE tensor (all its columns) -- the split-path and the narrow GEMM kernels, dtc_pack_cols, the
      latent, VAE-loss and PPO-heads kernels -- adds the largest |value| it writes to the tensor's slot (one atomic max per workgroup in
      its epilogue); consumers of the tensor -- on any stream that is ordered after the producer, as every reader of the data is -- read
      the slot.  Slots are zeroed by `reset()` at the start of a trainer phase (WeightImages.__enter__), so a buffer that is rewritten
      phase after phase does not carry an old maximum along.
    * static: tensors that do not change during an update (the rollout storage), computed once by `static()` before the lanes fork.
    * everything else (outputs of torch ops, tensors a caller of the C ABI brings along) comes without a slot: the library computes the
      amax of such an operand itself, right in front of the consumer (one zero-fill + one launch per call for all its slot-less
      operands, include/dtc_hip.h).
    A tensor is identified by its base address, width and row stride; a published or static slot covers every column block of it.
    One registry per device, one phase at a time: entering a WeightImages block zeroes the records of the previous one (a consumer that
    still needed them would scale by 2^141, overflow fp16 and return NaN -- loud, as every misuse of a record is)."""

    SLOTS = 512

    def __init__(self, device):
        self.device = device
        self.rec = int(lib().dtc_amax_record_bytes())       # a slot is a record of 16 words on 16 cache lines (csrc/s3_core.hpp)
        self.arena = torch.zeros(self.SLOTS * self.rec // 4, dtype=torch.int32, device=device)      # published slots (reset every phase)
        self.fixed = torch.zeros(64 * self.rec // 4, dtype=torch.int32, device=device)              # static slots
        self.index = {}            # key -> slot index in the arena
        self.fresh = {}            # keys whose slot is valid in this phase -> the tensor (held until the phase ends: a published
                                   # tensor's memory must not return to the caching allocator and come back as ANOTHER tensor with the
                                   # same key while its slot is live -- the recurrent trainers allocate their activations per step)
        self.static_index = {}     # base address -> index in `fixed`
        self.keep = {}

    def reset(self):
        used = len(self.index)                 # only the records handed out since the last reset can be non-zero
        if used:
            self.arena[:used * self.rec // 4].zero_()
        self.fresh.clear()
        self.index.clear()

    def _slot(self, key):
        i = self.index.get(key)
        if i is None:
            if len(self.index) >= self.SLOTS:          # checked BEFORE the key is registered: a failed request leaves no entry behind
                raise _ffi.DtcError("Amax: out of slots")
            i = self.index[key] = len(self.index)
        return self.arena.data_ptr() + self.rec * i

    @staticmethod
    def _key(t):
        return (t.data_ptr(), t.shape[1], t.stride(0))      # (a narrower view that starts at the same address is another tensor)

    def static(self, t):
        """(Re)compute the amax of a tensor that stays unchanged until the next call (on the current stream)."""
        if t.data_ptr() not in self.static_index and len(self.static_index) >= 64:
            raise _ffi.DtcError("Amax: out of static slots")
        i = self.static_index.setdefault(t.data_ptr(), len(self.static_index))
        self.keep[t.data_ptr()] = t
        p = self.fixed.data_ptr() + self.rec * i
        check(lib().dtc_amax(as_segmat(t.view(-1, t.shape[-1])), t.numel() // t.shape[-1], p, stream()), "dtc_amax")
        return p

    def out(self, t, col0, width):
        """Slot a producer of columns [col0, col0 + width) of `t` publishes into (None: not the whole tensor -- nothing published)."""
        if t is None or col0 != 0 or width != t.shape[1]:
            return None
        key = self._key(t)
        p = self._slot(key)                            # (may raise: the key must not become `fresh` without a slot)
        self.fresh[key] = t
        return p

    def of(self, t, kind=""):
        """Slot a consumer of (any column block of) `t` reads; None: the tensor has none (the library computes its amax)."""
        i = self.static_index.get(t.data_ptr())
        if i is not None:
            return self.fixed.data_ptr() + self.rec * i
        key = self._key(t)
        if key in self.fresh:
            if AMAX_CHECK:
                self._verify(t, self._slot(key))
            return self._slot(key)
        if AMAX_STATS is not None:
            k = ("no slot", kind, tuple(t.shape))
            AMAX_STATS[k] = AMAX_STATS.get(k, 0) + 1
        return None

    def _verify(self, t, p):
        torch.cuda.synchronize()
        w = (p - self.arena.data_ptr()) // 4
        have = max(int(v) & 0xffffffff for v in self.arena[w:w + self.rec // 4].tolist())
        true = t.detach().abs().max().view(torch.int32).item() if t.numel() else 0
        if have < true or (true > 0 and have > true + (12 << 23)):
            raise _ffi.DtcError(f"Amax: slot of a {tuple(t.shape)} tensor holds {have:#x}, the tensor's amax is {true:#x}")


_AMAX = {}         # device -> Amax: the registry the calls inside a WeightImages block (= one trainer phase) use


def amax_registry(device) -> Amax:
    a = _AMAX.get(device)
    if a is None:
        a = _AMAX[device] = Amax(device)
    return a


def amax_static(t):
    """Trainers: the amax of a tensor that stays unchanged until `amax_static_clear()` (the rollout storage during an update), computed
    once on the current stream -- call it before the compute lanes fork."""
    return amax_registry(t.device).static(t)


def amax_static_clear():
    for a in _AMAX.values():
        a.static_index.clear()
        a.keep.clear()


def _amax_in(t, kind=""):
    """amax slot of operand tensor `t` (None: no kernel published one and it is not static -- the library computes it)."""
    if _IMAGES is not None and _IMAGES.active:
        return amax_registry(t.device).of(t, kind)
    return None


def _amax_out(t, col0, width, rows=None):
    """`rows`: how many rows the producing kernel writes -- a kernel that covers only part of the tensor publishes the amax of that
    part, which a consumer of more rows must not scale by: no record is handed out then (the consumer computes the amax in its call)."""
    if _IMAGES is not None and _IMAGES.active and t is not None and (rows is None or rows == t.shape[0]):
        return amax_registry(t.device).out(t, col0, width)
    return None


def _sources(Xs):
    """The tensors behind a descriptor's segments (`_ffi.segmat` keeps them alive next to the raw pointers); a descriptor built any other
    way has none: its operands simply bring no amax record (the library computes it)."""
    keep = getattr(Xs, "_keep", None)
    srcs = list(keep[1]) if keep else []
    return srcs + [None] * (4 - len(srcs))


def _h2_operand(Xs, kind="fwd"):
    """Fill the amax slots of a row operand's segments (its source tensors ride along in the descriptor's keep-alive list)."""
    srcs = _sources(Xs)
    for i in range(Xs.nseg):
        Xs.seg[i].amax = _amax_in(srcs[i], kind) if srcs[i] is not None else None
    return Xs


def _pub(t):
    """Record a kernel that writes ALL of the 2-D tensor `t` publishes its amax into (None outside the fp16 path / a trainer phase)."""
    if not (SPLIT and H2) or t is None or t.dim() != 2 or _NOPUB_SMALL:
        return None
    return _amax_out(t, 0, t.shape[1])


def _h2_destination(dXs, rows=None):
    srcs = _sources(dXs)
    for i in range(dXs.nseg):
        s = dXs.seg[i]
        s.amax = _amax_out(srcs[i], s.col0, s.width, rows) if s.ptr else None
    return dXs


_PLANES = {}      # (device, stream, bytes) -> weight-image scratch of a split-path call


def _planes(W, rows, cols):
    """Scratch for the weight image of a [rows, cols] operand, one buffer per launch stream and size: calls on one
    stream are ordered, so the next call may overwrite it; the compute lanes of the trainer have their own."""
    n = int(lib().dtc_s3_planes_bytes(rows, cols))
    key = (W.device, stream(), n)
    buf = _PLANES.get(key)
    if buf is None:
        buf = _PLANES[key] = torch.empty((n + 7) // 8, dtype=torch.float64, device=W.device)
    return buf


class WeightImages:
    """Weight images (include/dtc_hip.h: DtcWimgJob) of all split-path layers of ONE trainer phase, built by one grouped launch
    at the start of the phase instead of one small launch in front of every GEMM (~29 per mini-batch of PPO.update).

        imgs = ops.WeightImages()
        with imgs:                     # start of an optimisation step, BEFORE the lanes fork: builds every image recorded so far
            ... forward / backward of the step (ops.linear_fwd / linear_dgrad / linear_fwd_mse) ...
        # the optimiser step follows outside

    The set is learnt: a split-path call inside the block that has no image yet gets a persistent buffer, builds its image
    itself this time (wimage_ready = 0) and is part of the grouped launch from the next block on.  Contract: the weights must
    not change between entering the block and the last call that uses them (the optimiser steps after the block; a caller that
    overwrites weights -- a test loading the oracle's -- does so between blocks), and the streams the calls run on must be
    ordered after the stream current at entry (the trainers fork their lanes after it and join them before the next block)."""

    def __init__(self):
        self.entries = {}          # key -> [image tensor, DtcWimgJob fields, built-in-this-block flag]
        self.jobs = None           # ctypes array of all entries (rebuilt when the set grows)
        self.keep = []
        self.active = False

    def __enter__(self):
        global _IMAGES
        self.prev, _IMAGES = _IMAGES, self
        self.active = True
        if SPLIT and H2:           # two-term fp16 path: a new phase -- the published amax slots start from zero
            for a in _AMAX.values():
                a.reset()
        if self.entries and SPLIT and WIMG:
            if self.jobs is None or sum(len(j) for j in self.jobs.values()) != len(self.entries):
                self.jobs = {}
                for rep in (False, True):                      # bf16 x 3 images and two-term fp16 images: one grouped launch each
                    es = [e for e in self.entries.values() if e[3] == rep]
                    if es:
                        arr = self.jobs[rep] = (_ffi.DtcWimgJob * len(es))()
                        for a, e in zip(arr, es):
                            a.W, a.img, a.seg, a.N, a.K, a.trans = e[1]
            for rep, arr in self.jobs.items():
                check((lib().dtc_h2_wimage_group if rep else lib().dtc_s3_wimage_group)(arr, len(arr), stream()), "dtc_s3_wimage_group")
            for e in self.entries.values():
                e[2] = True
        return self

    def __exit__(self, *exc):
        global _IMAGES
        _IMAGES = self.prev
        self.active = False
        for e in self.entries.values():
            e[2] = False
        return False

    def lookup(self, W, segs, N, K, trans, h2=False):
        """-> (image buffer, ready) for the call (W, operand segments, orientation, representation)."""
        key = (W.data_ptr(), N, K, trans, tuple((segs.seg[i].width, bool(segs.seg[i].ptr)) for i in range(segs.nseg)), h2)
        e = self.entries.get(key)
        if e is None:
            n = int(lib().dtc_s3_planes_bytes(K, N) if trans else lib().dtc_s3_planes_bytes(N, K))
            img = torch.zeros((n + 7) // 8, dtype=torch.float64, device=W.device)
            own = _ffi.DtcSegMat.from_buffer_copy(segs)           # the job keeps its own copy of the descriptor: the image
            own.idx = None                                        # builder reads the segment walk (widths, which destination
            for i in range(own.nseg):                             # blocks are NULL), never the operand: no address of the
                if own.seg[i].ptr:                                # first call's tensors survives in the copy
                    own.seg[i].ptr = _NOT_NULL
                own.seg[i].gather = 0
                own.seg[i].amax = None
            self.keep.append((W, own))
            e = self.entries[key] = [img, (cptr(W, f32), ptr(img), C.pointer(own), N, K, trans), False, h2]
        elif e[2] and WIMG_CHECK:
            self._verify(e, key)
        return e[0], int(e[2])

    def _verify(self, e, key):
        """DTC_WIMG_CHECK=1 (debug): the image built at block entry must still be the image of the weights this call sees --
        a caller that changed W inside the block (the contract above) is caught here instead of computing on stale planes."""
        fresh = torch.zeros_like(e[0])
        job = (_ffi.DtcWimgJob * 1)()
        job[0].W, _img, job[0].seg, job[0].N, job[0].K, job[0].trans = e[1]
        job[0].img = ptr(fresh)
        check((lib().dtc_h2_wimage_group if key[5] else lib().dtc_s3_wimage_group)(job, 1, stream()), "dtc_s3_wimage_group")
        n = int(lib().dtc_s3_planes_bytes(key[2], key[1]) if key[3] else lib().dtc_s3_planes_bytes(key[1], key[2]))
        if key[5]:
            # fp16 images: behind the chunks (8 KiB per 128-row tile and 16-k stage) and the 32 partial maxima of |W| the buffer holds the
            # call-private amax scratch of operands that arrive without a record -- not part of the image, a fresh build has zeros there
            rows, red = (key[2], key[1]) if key[3] else (key[1], key[2])
            n = min(n, -(-rows // 128) * -(-red // 16) * 8192 + 4 * 32)
        if not torch.equal(fresh.view(torch.uint8)[:n], e[0].view(torch.uint8)[:n]):
            raise _ffi.DtcError(f"WeightImages: weights of layer N={key[1]} K={key[2]} trans={key[3]} changed inside the block "
                                "(stale weight image)")


_IMAGES = None       # the WeightImages block the current calls run in, if any


def _wimage(W, segs, N, K, trans, h2=False):
    if _IMAGES is not None:
        return _IMAGES.lookup(W, segs, N, K, trans, h2)
    return (_planes(W, K, N) if trans else _planes(W, N, K)), 0


def relu_mask_ok(M, N):
    """Shapes for which a ReLU layer can record its output signs (dtc_linear_fwd_mask / dtc_linear_dgrad_mask)."""
    return M % 128 == 0 and (N % 128 == 0 or N == 64)


def relu_mask(M, N, device):
    return torch.empty(int(lib().dtc_relu_mask_elems(M, N)), dtype=torch.int16, device=device)


def pack_cols(X, dst, rows=None):
    """dst[rows, :X.cols] = the segments of DtcSegMat X side by side (gathered where asked)."""
    rows = dst.shape[0] if rows is None else rows
    check(lib().dtc_pack_cols(X, ptr(dst), dst.stride(0), rows, _amax_out(dst, 0, X.cols) if (SPLIT and H2 and rows == dst.shape[0]) else None,
                              stream()), "dtc_pack_cols")
    return dst


def linear_fwd(X, W, b, Y, act=None, M=None, mask=None, split=None):
    """Y = act(X W^T + b) on fp32 operands.  X: tensor or DtcSegMat; W [N,K]; Y [M,>=N] (row stride may exceed N).  `mask` (relu_mask
    buffer, act must be "relu"): also record the output signs for linear_dgrad(..., mask=).  (The trainers' wide stacks run on operand
    images: dtc_amd/h2i.py.)"""
    Xs = as_segmat(X)
    N, K = W.shape
    M = Y.shape[0] if M is None else M
    if (SPLIT if split is None else split) and (split or (N >= SPLIT_MIN_COLS and K >= SPLIT_MIN_RED)) and (mask is None or N % 128 == 0):
        img, ready = _wimage(W, Xs, N, K, 0, H2)
        if H2:
            check(lib().dtc_linear_fwd_h2(_h2_operand(Xs), cptr(W, f32), cptr(b, f32) if b is not None else None, ptr(Y), Y.stride(0),
                                          ptr(mask) if mask is not None else None, ptr(img), ready, _amax_out(Y, 0, N, M), M, N, K, ACT[act],
                                          stream()), "dtc_linear_fwd_h2")
            return Y
        check(lib().dtc_linear_fwd_s3(Xs, cptr(W, f32), cptr(b, f32) if b is not None else None, ptr(Y), Y.stride(0),
                                      ptr(mask) if mask is not None else None, ptr(img), ready, M, N, K, ACT[act], stream()),
              "dtc_linear_fwd_s3")
        return Y
    slot = _amax_out(Y, 0, N, M) if (SPLIT and H2) else None
    if slot is not None:          # a narrow layer inside a trainer phase of the fp16 path: its result's amax rides along for the consumers
        assert mask is None or act in ("relu", "crelu")
        check(lib().dtc_linear_fwd_amax(Xs, cptr(W, f32), cptr(b, f32) if b is not None else None, ptr(Y), Y.stride(0),
                                        ptr(mask) if mask is not None else None, slot, M, N, K, ACT[act], stream()), "dtc_linear_fwd_amax")
        return Y
    if mask is not None:
        assert act in ("relu", "crelu")
        check(lib().dtc_linear_fwd_mask(Xs, cptr(W, f32), cptr(b, f32) if b is not None else None, ptr(Y), Y.stride(0),
                                        ptr(mask), M, N, K, stream()), "dtc_linear_fwd_mask")
        return Y
    check(lib().dtc_linear_fwd(Xs, cptr(W, f32), cptr(b, f32) if b is not None else None, ptr(Y), Y.stride(0), M, N,
                               K, ACT[act], stream()), "dtc_linear_fwd")
    return Y


class FwdChain:
    """A fixed chain of forward layers marshalled ONCE (dtc_linear_fwd_list): `layers` = list of (X, W, b, Y, act) with X a
    tensor or DtcSegMat.  `run()` launches the whole chain with one FFI call.  Inputs that change between calls (the
    env's observation tensors) are re-pointed with `set_input(layer, segment, tensor)` -- the shape must stay the same."""

    def __init__(self, layers, M):
        self.M = M
        self.arr = (_ffi.DtcFwdLayer * len(layers))()
        self._keep = []
        for i, (X, W, b, Y, act) in enumerate(layers):
            Xs = as_segmat(X)
            N, K = W.shape
            a = self.arr[i]
            a.X = Xs
            a.W, a.b, a.Y, a.ldy = cptr(W, f32), (cptr(b, f32) if b is not None else None), ptr(Y), Y.stride(0)
            a.N, a.K, a.act = N, K, ACT[act]
            self._keep.append((Xs, W, b, Y))
        self._inputs = {}

    def set_input(self, layer, segment, t):
        s = self.arr[layer].X.seg[segment]
        if t.dtype != f32 or t.stride(1) != 1:
            t = t.contiguous().float()
        s.ptr, s.ld, s.rows = ptr(t), t.stride(0), t.shape[0]
        self._inputs[(layer, segment)] = t             # keep the tensor alive while the chain may run

    def run(self, stream_ptr=None):
        check(lib().dtc_linear_fwd_list(self.arr, len(self.arr), self.M, stream() if stream_ptr is None else stream_ptr),
              "dtc_linear_fwd_list")


def linear_dgrad(dZ, W, dX, Xsaved=None, act=None, M=None, mask=None, split=None):
    """dX = (dZ W) * act'(Xsaved) on fp32 operands; dX: tensor or DtcSegMat (destination).  `mask`: the sign record of the ReLU layer
    that produced Xsaved (then Xsaved itself is not read)."""
    dXs = as_segmat(dX)
    N, K = W.shape
    M = dZ.shape[0] if M is None else M
    if (SPLIT if split is None else split) and (split or (K >= SPLIT_MIN_COLS and N >= SPLIT_MIN_RED)) and (mask is None or K % 128 == 0):
        img, ready = _wimage(W, dXs, N, K, 1, H2)
        if H2:
            check(lib().dtc_linear_dgrad_h2(ptr(dZ), dZ.stride(0), _amax_in(dZ, "dgrad"), cptr(W, f32), _h2_destination(dXs, M),
                                            ptr(Xsaved) if mask is None else None, Xsaved.stride(0) if Xsaved is not None else 0,
                                            ptr(mask) if mask is not None else None, ptr(img), ready, M, N, K,
                                            ACT[act] if mask is None else ACT["relu"], stream()), "dtc_linear_dgrad_h2")
            return
        check(lib().dtc_linear_dgrad_s3(ptr(dZ), dZ.stride(0), cptr(W, f32), dXs, ptr(Xsaved) if mask is None else None,
                                        Xsaved.stride(0) if Xsaved is not None else 0, ptr(mask) if mask is not None else None,
                                        ptr(img), ready, M, N, K, ACT[act] if mask is None else ACT["relu"], stream()),
              "dtc_linear_dgrad_s3")
        return
    if SPLIT and H2:                  # a single whole-tensor destination publishes its amax from the narrow kernels as well (several
        if dXs.nseg == 1:             # blocks: they publish nothing, so no record may be handed out for them)
            _h2_destination(dXs, M)
        else:
            for i in range(dXs.nseg):
                dXs.seg[i].amax = None
    if mask is not None:
        assert act in ("relu", "crelu")
        check(lib().dtc_linear_dgrad_mask(ptr(dZ), dZ.stride(0), cptr(W, f32), dXs, ptr(mask), M, N, K, stream()),
              "dtc_linear_dgrad_mask")
        return
    check(lib().dtc_linear_dgrad(ptr(dZ), dZ.stride(0), cptr(W, f32), dXs, ptr(Xsaved),
                                 Xsaved.stride(0) if Xsaved is not None else 0, M, N, K, ACT[act], stream()),
          "dtc_linear_dgrad")


def wgrad_workspace_bytes(M, N, K) -> int:
    return int(lib().dtc_linear_wgrad_workspace(M, N, K))


def linear_wgrad(dZ, X, dW, db, workspace, M=None, stream_ptr=None, rows=None):
    """dW = dZ^T X, db = colsum(dZ).  X: tensor or DtcSegMat.  `stream_ptr`: raw HIP stream to launch on
    (default: torch's current stream).  `rows` (int64 device index, X a plain tensor): only these rows of BOTH operands enter the
    product -- every other row of dZ must be zero (the padding rows of a padded trajectory layout), so the result is the same sum
    without the work spent on zeros; taken on the split-precision path, ignored otherwise."""
    N, K = dW.shape
    M = dZ.shape[0] if M is None else M
    need = wgrad_workspace_bytes(M, N, K)
    if workspace.numel() * workspace.element_size() < need:
        raise _ffi.DtcError(f"wgrad workspace too small: {workspace.numel() * workspace.element_size()} < {need}")
    if (rows is not None and WGRAD_ROWS and SPLIT and isinstance(X, torch.Tensor) and N * K >= 128 * 128 and rows.numel() >= 1024
            and rows.numel() < M):
        check(lib().dtc_linear_wgrad_rows(ptr(dZ), dZ.stride(0), M, cptr(X, f32), X.stride(0), X.shape[0], cptr(rows, torch.int64),
                                          cptr(dW, f32), cptr(db, f32) if db is not None else None, ptr(workspace), rows.numel(),
                                          N, K, stream() if stream_ptr is None else stream_ptr), "dtc_linear_wgrad_rows")
        return
    Xs = as_segmat(X)
    check(lib().dtc_linear_wgrad(ptr(dZ), dZ.stride(0), Xs, cptr(dW, f32), cptr(db, f32) if db is not None else None,
                                 ptr(workspace), M, N, K, stream() if stream_ptr is None else stream_ptr),
          "dtc_linear_wgrad")


def _wgrad_jobs(jobs):
    """jobs: list of (dZ, X (tensor or DtcSegMat), dW, db or None) -> ctypes array + the objects it points into."""
    arr = (_ffi.DtcWgradJob * len(jobs))()
    keep = []
    for i, (dZ, X, dW, db) in enumerate(jobs):
        Xs = as_segmat(X)
        N, K = dW.shape
        arr[i].dZ, arr[i].lddz = ptr(dZ), dZ.stride(0)
        arr[i].X = Xs
        arr[i].dW = cptr(dW, f32)
        arr[i].db = cptr(db, f32) if db is not None else None
        arr[i].N, arr[i].K = N, K
        keep.append((dZ, Xs, dW, db))
    return arr, keep


def wgrad_group_workspace_bytes(jobs, M, split=None) -> int:
    arr, _ = _wgrad_jobs(jobs)
    s3 = SPLIT if split is None else split
    n = int((lib().dtc_wgrad_group_s3_workspace if s3 else lib().dtc_wgrad_group_workspace)(arr, len(jobs), M))
    if n < 0:
        raise _ffi.DtcError(f"dtc_wgrad_group_workspace failed: {lib().dtc_last_error().decode()}")
    return n


def wgrad_group(jobs, M, workspace, stream_ptr=None, split=None):
    """The weight gradients of several layers (one gradient bucket) in one partial launch + one reduce launch.
    jobs: list of (dZ [M,N], X tensor | DtcSegMat [M,K], dW [N,K], db [N] | None)."""
    arr, keep = _wgrad_jobs(jobs)
    s3 = SPLIT if split is None else split
    sp = stream() if stream_ptr is None else stream_ptr
    if s3 and H2:
        for a, (dZ, Xs, _dW, _db) in zip(arr, keep):
            a.dz_amax = _amax_in(dZ, "wgrad dZ")
            srcs = _sources(Xs)
            for i in range(Xs.nseg):
                a.X.seg[i].amax = _amax_in(srcs[i], "wgrad X") if srcs[i] is not None else None
        check(lib().dtc_wgrad_group_h2(arr, len(jobs), M, ptr(workspace), sp), "dtc_wgrad_group_h2")
        return keep
    check((lib().dtc_wgrad_group_s3 if s3 else lib().dtc_wgrad_group)(arr, len(jobs), M, ptr(workspace), sp),
          "dtc_wgrad_group_s3" if s3 else "dtc_wgrad_group")
    return keep


def mfma_sustained(device, random_operands: bool, launches=12, iters=2000, blocks=768, h2=None):
    """TFLOP/s of fp32-equivalent work the bare MFMA stream of the split kernels sustains on this chip for all-zero or random operand
    bits: the chip clocks to its power budget, so the two differ.  h2 (default: the active representation): the two-term fp16 stream
    (12 fp16 MFMAs per stage, fp16 FLOP / 3; dtc_probe_mfma_stream_h2) instead of the bf16 x 3 one (24 per stage, bf16 FLOP / 6)."""
    h2 = H2 if h2 is None else h2
    src = torch.randn(32768, device=device) if random_operands else torch.zeros(32768, device=device)
    ops_bits = src.half() if h2 else src.bfloat16()
    sink = torch.zeros(4, device=device)
    fn = lib().dtc_probe_mfma_stream_h2 if h2 else lib().dtc_probe_mfma_stream
    run = lambda: check(fn(ptr(ops_bits), blocks, iters, ptr(sink), stream()), "dtc_probe_mfma_stream")
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(launches):
        run()
    e1.record()
    torch.cuda.synchronize()
    per_stage, passes = (12, 3.0) if h2 else (24, 6.0)
    flop = float(launches) * blocks * 4 * iters * per_stage * 32768
    return flop / (e0.elapsed_time(e1) * 1e-3) / passes / 1e12


# ---------------------------------------------------------------- CE-net latent / losses / optimiser
# ---------------------------------------------------------------- device random draws of the update
def draw_seed() -> int:
    """A 62-bit seed from torch's CPU generator: reproducible under torch.manual_seed, no device synchronisation."""
    return int(torch.randint(0, 1 << 62, (1,), dtype=torch.int64).item())


def randn(shape, device, seed: int, offset: int = 0) -> torch.Tensor:
    out = torch.empty(shape, dtype=f32, device=device)
    check(lib().dtc_randn(ptr(out), out.numel(), seed, offset, stream()), "dtc_randn")
    return out


def randperm(n: int, device, seed: int) -> torch.Tensor:
    out = torch.empty(n, dtype=torch.int64, device=device)
    check(lib().dtc_randperm(ptr(out), n, seed, stream()), "dtc_randperm")
    return out


def workspace(nbytes: int, device) -> torch.Tensor:
    """8-byte aligned scratch of at least `nbytes` bytes."""
    return torch.empty(max(1, (int(nbytes) + 7) // 8), dtype=torch.float64, device=device)


def cenet_latent_fwd(mulv, eps, z, mask, info, ws, zmu_img=None):
    """`zmu_img` (h2i.HImage [B, 19]): also written by the launch -- the operand image of [z | mu[:, :3]]."""
    B = mulv.shape[0]
    assert zmu_img is None or (zmu_img.M, zmu_img.K) == (B, 19)
    check(lib().dtc_cenet_latent_fwd_img(cptr(mulv, f32), cptr(eps, f32), cptr(z, f32), cptr(mask, torch.uint8),
                                         cptr(info, torch.int32), ptr(ws), B, _pub(z), zmu_img.ptr() if zmu_img is not None else None,
                                         stream()), "dtc_cenet_latent_fwd")


def cenet_latent_bwd(dmulv, dz, eps, mulv, mask, info, ws):
    B = mulv.shape[0]
    check(lib().dtc_cenet_latent_bwd(cptr(dmulv, f32), cptr(dz, f32), cptr(eps, f32), cptr(mulv, f32),
                                     cptr(mask, torch.uint8), cptr(info, torch.int32), ptr(ws), B, _pub(dmulv), stream()),
          "dtc_cenet_latent_bwd")


def vae_loss(recons, hrecon, mulv, next_obs, priv, base_vel, idx, d_recons, d_hrecon, dmulv, losses, ws):
    B = recons.shape[0]
    check(lib().dtc_vae_loss(cptr(recons, f32), cptr(hrecon, f32), cptr(mulv, f32), cptr(next_obs, f32),
                             cptr(priv, f32), cptr(base_vel, f32), cptr(idx, torch.int64), cptr(d_recons, f32),
                             cptr(d_hrecon, f32), cptr(dmulv, f32), ptr(losses), ptr(ws), B, _pub(d_recons), stream()), "dtc_vae_loss")


def linear_fwd_mse(X, W, b, target, tcol0, tidx, dY, sq_part, M=None, split=None):
    """Output layer fused with its MSE loss: dY = 2/(M*N) * ((X W^T + b) - target[tidx, tcol0:tcol0+N]); sum of squared
    errors per workgroup into `sq_part` (float64, >= mse_parts(M, N) slots)."""
    Xs = as_segmat(X)
    N, K = W.shape
    M = dY.shape[0] if M is None else M
    s3 = (SPLIT if split is None else split) and (split or N >= SPLIT_MIN_COLS)
    n_part = int((lib().dtc_linear_fwd_mse_s3_parts if s3 else lib().dtc_linear_fwd_mse_parts)(M, N))
    if sq_part.numel() < n_part or sq_part.dtype != torch.float64:
        raise _ffi.DtcError(f"sq_part needs {n_part} float64 slots")
    args = (Xs, cptr(W, f32), cptr(b, f32) if b is not None else None, cptr(target, f32), target.stride(0), target.shape[0], tcol0,
            cptr(tidx, torch.int64), 2.0 / (M * N), ptr(dY), dY.stride(0), ptr(sq_part))
    if s3:
        img, ready = _wimage(W, Xs, N, K, 0, H2)
        if H2:
            _h2_operand(Xs)
            check(lib().dtc_linear_fwd_mse_h2(*args, ptr(img), ready, _amax_out(dY, 0, N, M), M, N, K, stream()), "dtc_linear_fwd_mse_h2")
            return n_part
        check(lib().dtc_linear_fwd_mse_s3(*args, ptr(img), ready, M, N, K, stream()), "dtc_linear_fwd_mse_s3")
    else:
        check(lib().dtc_linear_fwd_mse(*args, M, N, K, stream()), "dtc_linear_fwd_mse")
    return n_part


def vae_loss_fused(recons, mulv, next_obs, base_vel, idx, d_recons, dmulv, height_sq_part, n_height_part, losses, ws, drec_img=None):
    """`drec_img` (h2i.HImage [B, 53]): also written by the launch -- the operand image of d_recons."""
    B = recons.shape[0]
    assert drec_img is None or (drec_img.M, drec_img.K) == (B, d_recons.shape[1])
    check(lib().dtc_vae_loss_fused_img(cptr(recons, f32), cptr(mulv, f32), cptr(next_obs, f32), cptr(base_vel, f32),
                                       cptr(idx, torch.int64), cptr(d_recons, f32), cptr(dmulv, f32), ptr(height_sq_part),
                                       n_height_part, ptr(losses), ptr(ws), B, _pub(d_recons),
                                       drec_img.ptr() if drec_img is not None else None, stream()), "dtc_vae_loss_fused")


def ppo_heads_loss(Ha, Hc, Wa, ba, Wc, bc, act_prev, std, actions, old_logp, old_mu, old_sigma, advantages, returns, old_values,
                   idx, cfg, mean, value, dmean, dvalue, dHa, dHc, dstd, losses, lr, ws, imgs=None):
    """Output layers of actor and critic + dtc_ppo_loss + their data gradients in one launch (see dtc_hip.h).
    Ha / Hc [B,H] last hidden activations (post-activation, `act_prev`), dHa / dHc [B,H] receive their gradients.
    `imgs` = (dHa, dHc, dmean, dvalue) as h2i.HImage or None each: also written by the launch."""
    B, H = Ha.shape
    A = Wa.shape[0]
    ip = [None] * 4 if imgs is None else [im.ptr() if im is not None else None for im in imgs]
    if imgs is not None:
        for im, w in zip(imgs, (H, H, A, 1)):
            assert im is None or (im.M, im.K) == (B, w)
    check(lib().dtc_ppo_heads_loss_img(cptr(Ha, f32), Ha.stride(0), cptr(Hc, f32), Hc.stride(0), H, cptr(Wa, f32), cptr(ba, f32),
                                   cptr(Wc, f32), cptr(bc, f32), ACT[act_prev], ptr(std), cptr(actions, f32), cptr(old_logp, f32),
                                   cptr(old_mu, f32), cptr(old_sigma, f32), cptr(advantages, f32), cptr(returns, f32),
                                   cptr(old_values, f32), cptr(idx, torch.int64) if idx is not None else None, cfg,
                                   cptr(mean, f32), cptr(value, f32), cptr(dmean, f32), cptr(dvalue, f32), cptr(dHa, f32),
                                   dHa.stride(0) if dHa is not None else H, cptr(dHc, f32), dHc.stride(0) if dHc is not None else H, ptr(dstd), ptr(losses), ptr(lr), ptr(ws), B, A,
                                   _pub(dHa), _pub(dHc), _pub(dmean), _pub(dvalue), *ip, stream()), "dtc_ppo_heads_loss")


def ppo_loss(mean, std, value, actions, old_logp, old_mu, old_sigma, advantages, returns, old_values, idx, cfg,
             dmean, dvalue, dstd, losses, lr, ws):
    B, A = mean.shape
    check(lib().dtc_ppo_loss(cptr(mean, f32), ptr(std), cptr(value, f32), cptr(actions, f32), cptr(old_logp, f32),
                             cptr(old_mu, f32), cptr(old_sigma, f32), cptr(advantages, f32), cptr(returns, f32),
                             cptr(old_values, f32), cptr(idx, torch.int64) if idx is not None else None, cfg,
                             cptr(dmean, f32), cptr(dvalue, f32), ptr(dstd), ptr(losses), ptr(lr), ptr(ws), B, A,
                             stream()), "dtc_ppo_loss")


def gaussian_act(mean, std, noise, actions, logp, mu_out=None, sigma_out=None):
    B, A = mean.shape
    check(lib().dtc_gaussian_act(cptr(mean, f32), ptr(std), cptr(noise, f32), cptr(actions, f32), cptr(logp, f32),
                                 ptr(mu_out), ptr(sigma_out), B, A, stream()), "dtc_gaussian_act")


def bootstrap_probability(rewards) -> float:
    """actor_critic_decoder.py:404-407 (1 - tanh(std / mean), unbiased std); one launch + the reference's own `.item()`."""
    r = rewards.reshape(-1)
    if r.dtype != f32 or not r.is_contiguous():
        r = r.to(f32).contiguous()
    out = torch.empty(1, dtype=f32, device=r.device)
    check(lib().dtc_bootstrap_probability(cptr(r, f32), r.numel(), cptr(out, f32), stream()), "dtc_bootstrap_probability")
    return out.item()


def clip_adam(params, grads, exp_avg, exp_avg_sq, max_grad_norm, lr, beta1, beta2, eps, step, gnorm_out, ws):
    n = params.numel()
    check(lib().dtc_clip_adam(ptr(params), ptr(grads), ptr(exp_avg), ptr(exp_avg_sq), n, max_grad_norm, ptr(lr),
                              beta1, beta2, eps, step, ptr(gnorm_out), ptr(ws), stream()), "dtc_clip_adam")


def lr_adapt(kl_mean, lr, desired_kl, kl_out=None):
    """`kl_out` (1 float, optional): receives the KL the rule was evaluated on (the step's statistics row) -- same launch, no copy."""
    check(lib().dtc_lr_adapt(ptr(kl_mean), ptr(lr), desired_kl, ptr(kl_out) if kl_out is not None else None, stream()), "dtc_lr_adapt")


# ---------------------------------------------------------------- GRU
def gru_workspace_bytes(T, R, H) -> int:
    return int(lib().dtc_gru_workspace(T, R, H))


def gru_seq_check():
    """Raise if a persistent recurrence launch (csrc/gru_seq.hip) gave up at a barrier since the last check: more than two of them shared
    the device and their workgroups could not all be resident (two trainer processes on ONE GPU).  Call behind a device synchronisation."""
    if lib().dtc_gru_seq_status(1) == 1:
        raise _ffi.DtcError("a persistent GRU launch (dtc_gru_seq_fwd / _bwd) gave up at a barrier: its workgroups never met -- several "
                            "trainers share this device; set DTC_GRU_SEQ=0 (per-step launches) for that configuration")


def gru_seq_allow(on: bool):
    """Switch the persistent recurrence launches on / off for this process (dtc_set_gru_seq)."""
    lib().dtc_set_gru_seq(int(bool(on)))


def gru_fwd(gi, h0, W_hh, b_hh, hs_all, gates, hn, ws):
    """gi [T,R,3H], h0 [R,H] -> hs_all [T+1,R,H] (slot 0 = h0), gates [T,R,3H], hn [T,R,H]."""
    T, R, H3 = gi.shape
    check(lib().dtc_gru_fwd(cptr(gi, f32), cptr(h0, f32), cptr(W_hh, f32), cptr(b_hh, f32), cptr(hs_all, f32),
                            cptr(gates, f32), cptr(hn, f32), ptr(ws), T, R, H3 // 3, stream()), "dtc_gru_fwd")


def gru_bwd(dhs, hs_all, gates, hn, W_hh, dgi, dW_hh, db_hh, dh0, ws, rows=None):
    """`rows`: the valid (t, r) slots t * R + r of a padded trajectory batch (int64 device index): the W_hh weight gradient skips
    the padding slots (their gradients are zero).  dW_hh = db_hh = None: no W_hh weight gradient (see gru_dgh_all)."""
    T, R, H = dhs.shape
    use = rows is not None and WGRAD_ROWS and dW_hh is not None
    check(lib().dtc_gru_bwd(cptr(dhs, f32), cptr(hs_all, f32), cptr(gates, f32), cptr(hn, f32), cptr(W_hh, f32),
                            cptr(dgi, f32), cptr(dW_hh, f32) if dW_hh is not None else None,
                            cptr(db_hh, f32) if db_hh is not None else None, cptr(dh0, f32), ptr(ws),
                            cptr(rows, torch.int64) if use else None, rows.numel() if use else 0, T, R, H,
                            stream()), "dtc_gru_bwd")


def gru_fwd_multi(items):
    """`items`: one (gi, h0, W_hh, b_hh, hs_all, gates, hn, ws) tuple per recurrence, all of one shape -- dtc_gru_fwd_multi: with two
    items every time step is ONE launch for both (bit-identical to gru_fwd on each)."""
    T, R, H3 = items[0][0].shape
    assert all(tuple(it[0].shape) == (T, R, H3) for it in items)
    arr = (_ffi.DtcGruFwdItem * len(items))()
    for a, (gi, h0, W_hh, b_hh, hs_all, gates, hn, ws) in zip(arr, items):
        a.gi, a.h0, a.W_hh, a.b_hh = cptr(gi, f32), cptr(h0, f32), cptr(W_hh, f32), cptr(b_hh, f32)
        a.hs_all, a.gates, a.hn, a.workspace = cptr(hs_all, f32), cptr(gates, f32), cptr(hn, f32), ptr(ws)
    check(lib().dtc_gru_fwd_multi(arr, len(items), T, R, H3 // 3, stream()), "dtc_gru_fwd_multi")


def gru_bwd_multi(items):
    """`items`: one (dhs, hs_all, gates, hn, W_hh, dgi, dh0, ws) tuple per recurrence, all of one shape -- dtc_gru_bwd_multi (no W_hh
    weight gradient: see gru_dgh_all)."""
    T, R, H = items[0][0].shape
    assert all(tuple(it[0].shape) == (T, R, H) for it in items)
    arr = (_ffi.DtcGruBwdItem * len(items))()
    for a, (dhs, hs_all, gates, hn, W_hh, dgi, dh0, ws) in zip(arr, items):
        a.dhs, a.hs_all, a.gates, a.hn, a.W_hh = cptr(dhs, f32), cptr(hs_all, f32), cptr(gates, f32), cptr(hn, f32), cptr(W_hh, f32)
        a.dgi, a.dh0, a.workspace = cptr(dgi, f32), cptr(dh0, f32), ptr(ws)
    check(lib().dtc_gru_bwd_multi(arr, len(items), T, R, H, stream()), "dtc_gru_bwd_multi")


def gru_dgh_all(ws, T, R, H):
    """dgh_all [T * R, 3H] inside the workspace of a dtc_gru_bwd call (the gradient w.r.t. the recurrent pre-activations; valid once
    the call has run): the dZ operand of the W_hh weight gradient when the caller forms it itself (dW_hh = None)."""
    off = int(lib().dtc_gru_dgh_offset(T, R, H))
    return ws.view(torch.float32)[off // 4: off // 4 + T * R * 3 * H].view(T * R, 3 * H)


def lstm_workspace_bytes(T, R, H) -> int:
    return int(lib().dtc_lstm_workspace(T, R, H))


def lstm_fwd(gi, h0, c0, W_hh, b_hh, hs_all, cs_all, gates, ws):
    """gi [T,R,4H], h0 / c0 [R,H] -> hs_all / cs_all [T+1,R,H] (slot 0 = h0 / c0), gates [T,R,4H] (i, f, g, o)."""
    T, R, H4 = gi.shape
    check(lib().dtc_lstm_fwd(cptr(gi, f32), cptr(h0, f32), cptr(c0, f32), cptr(W_hh, f32), cptr(b_hh, f32), cptr(hs_all, f32),
                             cptr(cs_all, f32), cptr(gates, f32), ptr(ws), T, R, H4 // 4, stream()), "dtc_lstm_fwd")


def lstm_bwd(dhs, hs_all, cs_all, gates, W_hh, dgi, dW_hh, db_hh, dh0, dc0, ws):
    T, R, H = dhs.shape
    check(lib().dtc_lstm_bwd(cptr(dhs, f32), cptr(hs_all, f32), cptr(cs_all, f32), cptr(gates, f32), cptr(W_hh, f32),
                             cptr(dgi, f32), cptr(dW_hh, f32), cptr(db_hh, f32), cptr(dh0, f32), cptr(dc0, f32), ptr(ws), T, R, H,
                             stream()), "dtc_lstm_bwd")


def scatter_rows(src, idx, dst):
    """dst[idx[r]] = src[r] for 2-D fp32 tensors (rows of src.shape[1] floats)."""
    check(lib().dtc_scatter_rows(cptr(src, f32), cptr(idx, torch.int64), cptr(dst, f32), src.shape[0], src.shape[1],
                                 stream()), "dtc_scatter_rows")
