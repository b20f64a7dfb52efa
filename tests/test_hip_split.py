"""Split-precision GEMMs (csrc/gemm_s3.hip: every fp32 operand as three bf16 terms, six bf16 MFMA passes, fp32 accumulate)
against fp64, NEXT TO the single-pass fp32 MFMA kernels on the same inputs: the split path must be as accurate as the
fp32 path (its error against fp64 within a small factor of the fp32 kernel's own), for every operand form the layers
use -- segmented / gathered inputs, k tails, ragged N, bias + activation, the ReLU sign record, segmented / accumulating /
partly-NULL gradient destinations, the ELU derivative through the saved output."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _err(y, ref):
    return float((y.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))


def _act(v, act):
    if act == "relu":
        return torch.relu(v)
    if act == "elu":
        return torch.nn.functional.elu(v)
    return v


@pytest.mark.parametrize("M,N,K,act", [(1024, 512, 512, "relu"), (384, 512, 693, "relu"), (300, 693, 512, None), (1000, 256, 512, "elu"),
                                       (24576, 512, 512, "elu"), (130, 128, 265, "relu"), (4096, 140, 70, "elu")])
def test_forward_split_is_as_accurate_as_the_fp32_kernel(M, N, K, act):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + N + 3 * K)
    X = torch.randn(M, K, generator=g)
    X *= 10.0 ** torch.randint(-3, 3, (M, 1), generator=g).float()            # rows of very different magnitude
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = _act(X.double() @ W.double().T + b.double(), act)
    Xd, Wd, bd = X.to(DEV), W.to(DEV), b.to(DEV)
    y32 = torch.full((M, N), float("nan"), device=DEV)
    ys3 = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(Xd, Wd, bd, y32, act, split=False)
    ops.linear_fwd(Xd, Wd, bd, ys3, act, split=True)
    e32, es3 = _err(y32, ref), _err(ys3, ref)
    print(f"fwd {M}x{N}x{K}: fp32 MFMA err {e32:.2e}, split err {es3:.2e}")
    assert es3 <= 2.0 * e32 + 2e-7, (e32, es3)
    # element-wise: every output within fp32-GEMM distance of the fp32 kernel's
    scale = (X.abs().double() @ W.abs().double().T + b.abs().double()).to(DEV)
    assert float(((ys3 - y32).abs().double() / scale).max()) <= 1e-6


def test_forward_split_segments_gather_and_sign_record():
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(5)
    B, R = 1536, 4000
    obs, z, mulv, lt = (torch.randn(R, 53, generator=g), torch.randn(B, 16, generator=g), torch.randn(B, 35, generator=g),
                        torch.randn(B, 512, generator=g))
    idx = torch.randint(0, R, (B,), generator=g)
    W = torch.randn(512, 584, generator=g) / 24.0
    b = torch.randn(512, generator=g)
    Xfull = torch.cat([obs[idx], z, mulv[:, :3], lt], dim=1)
    ref = torch.relu(Xfull.double() @ W.double().T + b.double())
    d = lambda t: t.to(DEV)
    obs_d, z_d, mulv_d, lt_d, idx_d = d(obs), d(z), d(mulv), d(lt), d(idx)
    X = _ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(z_d, 0, 16), _ffi.seg(mulv_d, 0, 3), _ffi.seg(lt_d, 0, 512)], idx_d)
    y32, ys3 = torch.empty(B, 512, device=DEV), torch.empty(B, 512, device=DEV)
    m32, ms3 = ops.relu_mask(B, 512, DEV), ops.relu_mask(B, 512, DEV)
    ops.linear_fwd(X, d(W), d(b), y32, "relu", M=B, mask=m32, split=False)
    ops.linear_fwd(X, d(W), d(b), ys3, "relu", M=B, mask=ms3, split=True)
    assert _err(ys3, ref) <= 2.0 * _err(y32, ref) + 2e-7
    # the sign record describes the split path's OWN output exactly (knife edges may differ from the fp32 kernel's)
    pos = (ys3 > 0).cpu().view(B // 32, 4, 2, 4, 512).permute(0, 2, 1, 3, 4).reshape(B // 32, 2, 16, 512).to(torch.int32)
    want = (pos << torch.arange(16, dtype=torch.int32).view(1, 1, 16, 1)).sum(dim=2).reshape(-1, 512)
    assert torch.equal(ms3.cpu().view(-1, 512).to(torch.int32) & 0xFFFF, want)
    assert float(((ys3 > 0) != (y32 > 0)).float().mean()) < 1e-4


@pytest.mark.parametrize("M,N,K,act", [(1024, 512, 512, "relu"), (384, 693, 512, "relu"), (512, 256, 512, "elu"), (300, 128, 256, "elu"),
                                       (24576, 512, 512, "relu")])
def test_data_gradient_split_is_as_accurate_as_the_fp32_kernel(M, N, K, act):
    """dX [M, K] = (dZ [M, N] W [N, K]) * act'(Xs)."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(2 * M + N + K)
    dZ = torch.randn(M, N, generator=g) * 10.0 ** torch.randint(-6, 0, (M, 1), generator=g).float()
    W = torch.randn(N, K, generator=g) / N ** 0.5
    Xs = _act(torch.randn(M, K, generator=g), act)
    ref = dZ.double() @ W.double()
    ref = ref * (Xs > 0) if act == "relu" else torch.where(Xs > 0, ref, ref * (Xs.double() + 1.0))
    d32 = torch.full((M, K), float("nan"), device=DEV)
    ds3 = torch.full((M, K), float("nan"), device=DEV)
    ops.linear_dgrad(dZ.to(DEV), W.to(DEV), d32, Xs.to(DEV), act, split=False)
    ops.linear_dgrad(dZ.to(DEV), W.to(DEV), ds3, Xs.to(DEV), act, split=True)
    e32, es3 = _err(d32, ref), _err(ds3, ref)
    print(f"dgrad {M}x{N}x{K}: fp32 MFMA err {e32:.2e}, split err {es3:.2e}")
    assert es3 <= 2.0 * e32 + 2e-7, (e32, es3)
    scale = (dZ.abs().double() @ W.abs().double()).to(DEV) + 1e-30
    assert float(((ds3 - d32).abs().double() / scale).max()) <= 1e-6


def test_data_gradient_split_destinations_and_sign_record():
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(8)
    B = 384
    dZ = torch.randn(B, 512, generator=g)
    Wa = torch.randn(512, 584, generator=g) / 22.0
    full = dZ.double() @ Wa.double()
    dz, dmulv, dlt0 = torch.zeros(B, 16, device=DEV), torch.zeros(B, 35, device=DEV), torch.randn(B, 512, generator=g)
    dlt = dlt0.to(DEV)
    dst = _ffi.segmat([_ffi.seg(None, 0, 53), _ffi.seg(dz, 0, 16), _ffi.seg(dmulv, 0, 3), _ffi.seg(dlt, 0, 512, accumulate=True)])
    ops.linear_dgrad(dZ.to(DEV), Wa.to(DEV), dst, split=True)
    np.testing.assert_allclose(dz.cpu().numpy(), full[:, 53:69].numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dmulv.cpu().numpy()[:, :3], full[:, 69:72].numpy(), rtol=2e-5, atol=2e-5)
    assert float(dmulv[:, 3:].abs().max()) == 0.0
    np.testing.assert_allclose(dlt.cpu().numpy(), (dlt0.double() + full[:, 72:]).numpy(), rtol=2e-5, atol=2e-5)
    # ReLU derivative from the sign record == from the saved activation (bitwise, both on the split path)
    M, N, K = 1024, 512, 512
    X = torch.randn(M, 300, generator=g).to(DEV)
    W1 = (torch.randn(K, 300, generator=g) / 17.0).to(DEV)
    Y = torch.empty(M, K, device=DEV)
    mask = ops.relu_mask(M, K, DEV)
    ops.linear_fwd(X, W1, None, Y, "relu", mask=mask, split=True)
    dZ2 = torch.randn(M, N, generator=g).to(DEV)
    W2 = (torch.randn(N, K, generator=g) / 22.0).to(DEV)
    a, b = torch.empty(M, K, device=DEV), torch.empty(M, K, device=DEV)
    ops.linear_dgrad(dZ2, W2, a, Y, "relu", split=True)
    ops.linear_dgrad(dZ2, W2, b, Y, "relu", mask=mask, split=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("M", [1536, 5000, 24576])
def test_weight_gradients_split_are_as_accurate_as_the_fp32_kernels(M):
    """dtc_wgrad_group_s3 on the layers of a bucket (wide, narrow, 1-row, segmented + gathered, ragged N / K, M not a
    multiple of the stage) against fp64, next to dtc_wgrad_group; deterministic (second call bit-identical)."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(100 + M)
    R = M + 700
    d = lambda t: t.to(DEV)
    obs_all, priv, bv = torch.randn(R, 53, generator=g), torch.randn(R, 1389, generator=g), torch.randn(R, 3, generator=g)
    idx = torch.randperm(R, generator=g)[:M]
    obs_d, priv_d, bv_d, idx_d = d(obs_all), d(priv), d(bv), d(idx)
    shapes = [(512, 752), (512, 512), (256, 512), (128, 265), (12, 128), (1, 128), (35, 64), (693, 512), (64, 531)]
    jobs32, jobs3, refs = [], [], []
    for li, (N, K) in enumerate(shapes):
        dZ = torch.randn(M, N, generator=g) / M ** 0.5 * 10.0 ** torch.randint(-4, 1, (M, 1), generator=g).float()
        if li == 0:
            Xh = torch.cat([obs_all[idx], bv[idx], priv[idx, 693:]], dim=1)
            mk = lambda: _ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(bv_d, 0, 3, gather=True),
                                      _ffi.seg(priv_d, 693, 696, gather=True)], idx_d)
        elif li == 8:                               # CE-net decoder input: [z 16 | mu 3 | l_t 512], plain segments
            zz, mm, ll = torch.randn(M, 16, generator=g), torch.randn(M, 35, generator=g), torch.randn(M, 512, generator=g)
            Xh = torch.cat([zz, mm[:, :3], ll], dim=1)
            zd, md, ld_ = d(zz), d(mm), d(ll)
            mk = lambda zd=zd, md=md, ld_=ld_: _ffi.segmat([_ffi.seg(zd, 0, 16), _ffi.seg(md, 0, 3), _ffi.seg(ld_, 0, 512)])
        else:
            Xh = torch.randn(M, K, generator=g)
            Xd = d(Xh)
            mk = lambda Xd=Xd: Xd
        dZd = d(dZ)
        has_b = li != 3
        jobs32.append((dZd, mk(), torch.full((N, K), float("nan"), device=DEV), torch.full((N,), float("nan"), device=DEV) if has_b else None))
        jobs3.append((dZd, mk(), torch.full((N, K), float("nan"), device=DEV), torch.full((N,), float("nan"), device=DEV) if has_b else None))
        refs.append((dZ.double().t() @ Xh.double(), dZ.double().sum(0), dZ.abs().double().t() @ Xh.abs().double()))
    ws32 = ops.workspace(ops.wgrad_group_workspace_bytes(jobs32, M, split=False), DEV)
    ws3 = ops.workspace(ops.wgrad_group_workspace_bytes(jobs3, M, split=True), DEV)
    ops.wgrad_group(jobs32, M, ws32, split=False)
    ops.wgrad_group(jobs3, M, ws3, split=True)
    for li, ((_, _, w32, b32), (_, _, w3, b3), (rw, rb, sc)) in enumerate(zip(jobs32, jobs3, refs)):
        assert torch.isfinite(w3).all(), li
        e32, e3 = _err(w32, rw), _err(w3, rw)
        print(f"wgrad M={M} {tuple(w3.shape)}: fp32 MFMA err {e32:.2e}, split err {e3:.2e}")
        assert e3 <= 2.0 * e32 + 2e-7, (li, e32, e3)
        assert float(((w3 - w32).abs().double().cpu() / (sc + 1e-30)).max()) <= 1e-6, li
        if b3 is not None:
            np.testing.assert_allclose(b3.cpu().numpy(), rb.numpy(), rtol=3e-5, atol=1e-6 * float(rb.abs().max()) + 1e-9)
    first = [(j[2].clone(), None if j[3] is None else j[3].clone()) for j in jobs3]
    ops.wgrad_group(jobs3, M, ws3, split=True)
    for (dZ, X, dW, db), (w0, b0) in zip(jobs3, first):
        assert torch.equal(dW, w0) and (db is None or torch.equal(db, b0))


def test_fused_mse_output_layer_split_vs_fp32():
    from dtc_amd import ops
    g = torch.Generator().manual_seed(3)
    M, N, K, R = 1536, 693, 512, 3000
    X = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / 22.0).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    tgt = torch.randn(R, 1389, generator=g).to(DEV)
    tidx = torch.randint(0, R, (M,), generator=g).to(DEV)
    outs = []
    for split in (False, True):
        dY = torch.full((M, N), float("nan"), device=DEV)
        part = torch.zeros(4096, dtype=torch.float64, device=DEV)
        n = ops.linear_fwd_mse(X, W, b, tgt, 696, tidx, dY, part, split=split)
        outs.append((dY, float(part[:n].sum())))
    e = (X.double() @ W.double().T + b.double()) - tgt[tidx][:, 696:].double()
    ref_dy, ref_sq = e * (2.0 / (M * N)), float((e * e).sum())
    for dY, sq in outs:
        assert abs(sq - ref_sq) <= 1e-6 * ref_sq
        assert _err(dY, ref_dy.cpu()) <= 2e-6
    assert _err(outs[1][0], ref_dy.cpu()) <= 2.0 * _err(outs[0][0], ref_dy.cpu()) + 2e-7


def test_grouped_weight_images_equal_per_call_images_bitwise():
    """ops.WeightImages: the images of a block's layers built by ONE launch at its start (second block on) give the same bits as the
    per-call images -- segmented / gathered forward operand with k tails, partly-NULL segmented gradient destination (column
    skip), fused MSE output layer, more jobs than one grouped launch holds -- and follow the weights when they change between blocks."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(11)
    B, R = 640, 2000
    obs, z, lt = torch.randn(R, 53, generator=g).to(DEV), torch.randn(B, 19, generator=g).to(DEV), torch.randn(B, 512, generator=g).to(DEV)
    idx = torch.randint(0, R, (B,), generator=g).to(DEV)
    Wa = (torch.randn(512, 584, generator=g) / 24.0).to(DEV)
    ba = torch.randn(512, generator=g).to(DEV)
    Ws = [(torch.randn(256, 512, generator=g) / 22.0).to(DEV) for _ in range(26)]           # > one launch's 24 jobs together with the rest
    Wo = (torch.randn(693, 256, generator=g) / 16.0).to(DEV)
    bo = torch.randn(693, generator=g).to(DEV)
    tgt = torch.randn(R, 1389, generator=g).to(DEV)
    dZ = torch.randn(B, 512, generator=g).to(DEV)

    def run():
        Xs = _ffi.segmat([_ffi.seg(obs, 0, 53, gather=True), _ffi.seg(z, 0, 19), _ffi.seg(lt, 0, 512)], idx)
        Y = torch.empty(B, 512, device=DEV)
        ops.linear_fwd(Xs, Wa, ba, Y, "elu", M=B, split=True)
        hs = []
        for W in Ws:
            h = torch.empty(B, 256, device=DEV)
            ops.linear_fwd(Y, W, None, h, "relu", split=True)
            hs.append(h)
        dY = torch.empty(B, 693, device=DEV)
        part = torch.zeros(4096, dtype=torch.float64, device=DEV)
        n = ops.linear_fwd_mse(hs[0], Wo, bo, tgt, 696, idx, dY, part, split=True)
        dz, dlt = torch.zeros(B, 19, device=DEV), torch.ones(B, 512, device=DEV)
        dst = _ffi.segmat([_ffi.seg(None, 0, 53), _ffi.seg(dz, 0, 19), _ffi.seg(dlt, 0, 512, accumulate=True)])
        ops.linear_dgrad(dZ, Wa, dst, split=True)
        dh = torch.empty(B, 512, device=DEV)
        ops.linear_dgrad(hs[1], Ws[1], dh, Y, "elu", split=True)
        return [Y, dY, part[:n].clone(), dz, dlt, dh] + hs

    ref = run()
    imgs = ops.WeightImages()
    for block in range(3):
        with imgs:
            out = run()
        assert len(imgs.entries) == 1 + 26 + 1 + 2
        for a, b in zip(ref, out):
            assert torch.equal(a, b), block
    Wa.mul_(1.5)                                   # the optimiser's step between two blocks
    Ws[1].add_(0.01)
    ref2 = run()
    assert not torch.equal(ref2[0], ref[0])
    with imgs:
        out2 = run()
    for a, b in zip(ref2, out2):
        assert torch.equal(a, b)
    out3 = run()                                   # outside a block: per-call images again
    for a, b in zip(ref2, out3):
        assert torch.equal(a, b)


def test_weight_gradient_over_a_row_map_skips_zero_rows():
    """ops.linear_wgrad(rows=...): only the listed rows of both operands enter the product (the recurrent trainers' valid slots of a
    padded trajectory batch; every other row of dZ is zero).  Same result as the product over all rows -- against fp64 and next to
    it --, also when the rows the map leaves out hold garbage in X."""
    from dtc_amd import ops
    if not ops.SPLIT:
        pytest.skip("row-mapped weight gradients exist on the split-precision path only (DTC_GEMM_SPLIT=0: the padded product runs)")
    g = torch.Generator().manual_seed(21)
    Mp, N, K = 6000, 1536, 512
    valid = torch.randperm(Mp, generator=g)[:4100].sort().values
    dZ = torch.zeros(Mp, N)
    dZ[valid] = torch.randn(valid.numel(), N, generator=g)
    X = torch.randn(Mp, K, generator=g)
    ref_w, ref_b = dZ.double().T @ X.double(), dZ.double().sum(0)
    Xg = X.clone()
    pad = torch.ones(Mp, dtype=torch.bool)
    pad[valid] = False
    Xg[pad] = float("nan")                                  # rows outside the map are never read
    d = lambda t: t.to(DEV)
    outs = []
    for rows, Xin in ((None, X), (valid, Xg)):
        dW, db = torch.full((N, K), float("nan"), device=DEV), torch.full((N,), float("nan"), device=DEV)
        ws = ops.workspace(ops.wgrad_workspace_bytes(Mp, N, K), DEV)
        ops.linear_wgrad(d(dZ), d(Xin), dW, db, ws, rows=d(rows) if rows is not None else None)
        outs.append((dW, db))
    for dW, db in outs:
        assert _err(dW, ref_w) <= 2e-6 and _err(db, ref_b) <= 2e-6
    assert _err(outs[1][0], ref_w) <= 2.0 * _err(outs[0][0], ref_w) + 2e-7


def test_split_timing_report():
    """Not an assertion on speed: prints the per-launch time of both paths on the bench's 512-wide layer."""
    from dtc_amd import ops
    M, N, K = 24576, 512, 512
    X, W, b = torch.randn(M, K, device=DEV), torch.randn(N, K, device=DEV) / 22.0, torch.randn(N, device=DEV)
    Y, dX = torch.empty(M, N, device=DEV), torch.empty(M, K, device=DEV)
    for split in (False, True):
        for name, fn in (("fwd", lambda: ops.linear_fwd(X, W, b, Y, "relu", split=split)),
                         ("dgrad", lambda: ops.linear_dgrad(Y, W, dX, X, "relu", split=split))):
            for _ in range(3):
                fn()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 50.0
            print(f"{name} split={split}: {us:.1f} us = {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s (fp32-equivalent)")
    # row alignment of the input: K = 693 columns out of rows of 693 (2772 bytes: not 16-byte aligned) vs 696 floats, plain and gathered
    from dtc_amd import _ffi
    K = 693
    W = torch.randn(N, K, device=DEV) / 26.0
    idx = torch.randperm(98304, device=DEV)[:M].contiguous()
    for ld in (693, 696):
        big = torch.randn(98304, ld, device=DEV)
        for gather in (False, True):
            Xs = _ffi.segmat([_ffi.seg(big, 0, K, gather=gather)], idx if gather else None)
            for split in (False, True):
                for _ in range(3):
                    ops.linear_fwd(Xs, W, b, Y, "relu", M=M, split=split)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ops.linear_fwd(Xs, W, b, Y, "relu", M=M, split=split)
                e1.record()
                torch.cuda.synchronize()
                us = e0.elapsed_time(e1) * 50.0
                print(f"fwd 24576x512x693 row stride {ld} gather={gather} split={split}: {us:.1f} us = {2.0 * M * N * K / us / 1e6:.1f} TFLOP/s")
