"""Block-scaled two-term fp16 operand images (include/dtc_hip.h, csrc/h2i_core.hpp, round 5) against fp64: pack / unpack, the
image-operand forward / data-gradient / MSE kernels and the grouped weight gradients -- the products of the nn.Linear stacks of
rsl_rl/rsl_rl/modules/actor_critic_decoder.py:98-188, 323-349.  What the per-row exponents promise is tested per ROW: every row of a
result is as accurate relative to ITSELF as an fp32 dot product, whatever magnitudes the other rows of the tensor have, and a
non-finite element poisons only the outputs that depend on it (the reference's behaviour: ppo.py:137-155 confines a diverged env)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROW_TOL = 2e-6          # per-row relative error bound (row error / row magnitude), measured ~3e-7


@pytest.fixture(autouse=True, params=["rows64_default", "rows128_only"])
def tile_rows(request):
    """Every test runs twice: small launches on 64-row tiles (the default for launches of at most one 128 x 128 tile per CU) and
    everything on 128-row tiles -- the shapes of this file are small enough that the default alone would never reach the latter."""
    from dtc_amd import _ffi
    _ffi.lib().dtc_h2i_rows64_max(-1 if request.param == "rows64_default" else 0)
    yield request.param
    _ffi.lib().dtc_h2i_rows64_max(-1)


def _act(v, act):
    return torch.relu(v) if act == "relu" else torch.nn.functional.elu(v) if act == "elu" else v


def _row_err(y, ref):
    """max over rows of (largest element error of the row / largest |element| of the row); all-zero reference rows must be exact"""
    y, ref = y.double().cpu(), ref.double().cpu()
    num = (y - ref).abs().amax(dim=1)
    den = ref.abs().amax(dim=1)
    zero = den == 0
    assert float(num[zero].max() if zero.any() else 0.0) == 0.0
    return float((num[~zero] / den[~zero]).max()) if (~zero).any() else 0.0


def _rows(M, K, g, span=12, zero_frac=0.0):
    """rows log-uniform over 10^-span .. 1 of the largest, a fraction of them exactly zero"""
    X = torch.randn(M, K, generator=g) * 10.0 ** (-span * torch.rand(M, 1, generator=g))
    if zero_frac:
        X[torch.rand(M, generator=g) < zero_frac] = 0.0
    return X


@pytest.mark.parametrize("M,K", [(128, 16), (300, 693), (4096, 512), (1, 5), (257, 1389)])
def test_pack_round_trip_per_row(M, K):
    from dtc_amd import h2i
    g = torch.Generator().manual_seed(M + K)
    X = _rows(M, K + 3, g, zero_frac=0.1).to(DEV)[:, :K]                       # row stride != K
    img = h2i.HImage.from_tensor(X)
    got = img.to_tensor()
    # every element within 2^-21 of its row block's largest element (22 significant bits)
    Xp = torch.zeros(M, -(-K // 128) * 128, device=DEV)
    Xp[:, :K] = X
    blk = Xp.view(M, -1, 128).abs().amax(dim=2, keepdim=True).expand(-1, -1, 128).reshape(M, -1)[:, :K]
    assert bool(((got - X).abs() <= blk * 2.0 ** -21).all())
    ex = img.exps()
    rowmax = Xp.view(M, -1, 128).abs().amax(dim=2)                            # [M, kb]
    e = ex.permute(0, 2, 1).reshape(-1, ex.shape[1])[:M]                      # [M, kb]
    assert bool((e[rowmax == 0] == 0x7fff).all())
    live = rowmax > 0
    scaled = rowmax[live] * torch.exp2(e[live].float())
    assert bool(((scaled >= 2.0 ** 14) & (scaled < 2.0 ** 15)).all())


def test_pack_gathered_segments():
    from dtc_amd import h2i
    from dtc_amd._ffi import seg, segmat
    g = torch.Generator().manual_seed(5)
    A, Bm, Cm = (torch.randn(900, w, generator=g).to(DEV) for w in (53, 3, 1389))
    idx = torch.randint(0, 900, (700,), generator=g).to(DEV)
    Xs = segmat([seg(A, 0, 53, gather=True), seg(Bm, 0, 3, gather=True), seg(Cm, 693, 696, gather=True)], idx)
    img = h2i.HImage(700, 752, DEV).pack(Xs)
    want = torch.cat([A[idx], Bm[idx], Cm[idx][:, 693:1389]], dim=1)
    got = img.to_tensor()
    assert float((got - want).abs().max()) <= 2.0 ** -21 * float(want.abs().max())
    assert _row_err(got, want) < 1e-6


@pytest.mark.parametrize("M,N,K,act", [(1024, 512, 512, "relu"), (384, 512, 693, "relu"), (300, 693, 512, None), (1000, 256, 512, "elu"),
                                       (4096, 512, 752, "elu"), (130, 128, 265, "relu"), (200, 140, 70, "elu"), (24576, 512, 512, "relu"),
                                       (1024, 64, 531, "relu"), (256, 128, 64, "relu"), (384, 53, 128, None), (512, 35, 64, None), (640, 64, 128, None)])
def test_forward_per_row_accuracy_and_image_result(M, N, K, act):
    from dtc_amd import h2i, ops
    g = torch.Generator().manual_seed(M + N + 3 * K)
    X = _rows(M, K, g, span=12, zero_frac=0.05)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.zeros(N) if M > 4096 else torch.randn(N, generator=g) * 1e-13    # (a bias would swamp the small rows' products)
    ref = _act(X.double() @ W.double().T + b.double(), act)
    Xd, Wd, bd = X.to(DEV), W.to(DEV), b.to(DEV)
    Y = torch.full((M, N), float("nan"), device=DEV)
    Yimg = h2i.HImage(M, N, DEV)
    mask = ops.relu_mask(M, N, DEV) if (act == "relu" and ops.relu_mask_ok(M, N)) else None      # N % 128 == 0 or N == 64
    h2i.linear_fwd(h2i.HImage.from_tensor(Xd), Wd, bd, Y, Yimg, act, mask=mask)
    err = _row_err(Y, ref)
    print(f"fwd {M}x{N}x{K}: per-row err {err:.2e}")
    assert err < ROW_TOL
    if mask is not None:                                   # the sign record = the signs of the result (ReLU output > 0)
        assert torch.equal(h2i.unpack_sign_record(mask, M, N), Y > 0)
    dec = Yimg.to_tensor()
    blk = torch.zeros(M, -(-N // 128) * 128, device=DEV)
    blk[:, :N] = Y.abs()
    blk = blk.view(M, -1, 128).amax(dim=2, keepdim=True).expand(-1, -1, 128).reshape(M, -1)[:, :N]
    assert bool(((dec - Y).abs() <= blk * 2.0 ** -21).all())
    # image only (no fp32 result) gives the same image
    Yimg2 = h2i.HImage(M, N, DEV)
    h2i.linear_fwd(h2i.HImage.from_tensor(Xd), Wd, bd, None, Yimg2, act)
    assert torch.equal(Yimg2.buf, Yimg.buf)


def test_forward_two_operand_images_and_column_map():
    """the actor's first layer: [l_t image | packed narrow block image] against W's columns [72:584 | 0:72]"""
    from dtc_amd import h2i
    g = torch.Generator().manual_seed(11)
    M = 640
    lt, nb = torch.randn(M, 512, generator=g), torch.randn(M, 72, generator=g) * 3
    W, b = torch.randn(512, 584, generator=g) / 24, torch.randn(512, generator=g)
    ref = torch.nn.functional.elu(torch.cat([nb, lt], 1).double() @ W.double().T + b.double())
    Y = torch.empty(M, 512, device=DEV)
    h2i.linear_fwd([h2i.HImage.from_tensor(lt.to(DEV)), h2i.HImage.from_tensor(nb.to(DEV))], W.to(DEV), b.to(DEV), Y, None, "elu", cols=[72, 0])
    assert _row_err(Y, ref) < ROW_TOL


@pytest.mark.parametrize("M,N,K", [(512, 693, 512), (300, 693, 512)])
def test_fused_mse_layer(M, N, K):
    from dtc_amd import h2i
    g = torch.Generator().manual_seed(M)
    X, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    T = torch.randn(2 * M, 1389, generator=g)
    idx = torch.randint(0, 2 * M, (M,), generator=g)
    e = (X.double() @ W.double().T + b.double()) - T[idx][:, 696:696 + N].double()
    dY = torch.empty(M, N, device=DEV)
    dYimg = h2i.HImage(M, N, DEV)
    part = torch.zeros(h2i.mse_parts(M, N), dtype=torch.float64, device=DEV)
    n = h2i.linear_fwd_mse(h2i.HImage.from_tensor(X.to(DEV)), W.to(DEV), b.to(DEV), T.to(DEV), 696, idx.to(DEV), dY, dYimg, part)
    assert abs(float(part[:n].sum()) - float((e * e).sum())) <= 1e-6 * float((e * e).sum())
    want = e * (2.0 / (M * N))
    assert _row_err(dY, want) < ROW_TOL
    assert float((dYimg.to_tensor() - dY).abs().max()) <= 2.0 ** -21 * float(dY.abs().max())


@pytest.mark.parametrize("M,N,K,mode", [(1024, 512, 512, "mask"), (640, 256, 512, "elu"), (300, 693, 512, "none"), (512, 512, 693, "none"),
                                        (384, 128, 256, "elu"), (24576, 512, 512, "mask"), (1024, 128, 64, "mask"), (512, 53, 128, "mask"),
                                        (640, 35, 64, "none"), (256, 64, 531, "none"), (384, 64, 128, "mask")])
def test_dgrad_heavy_tailed_rows(M, N, K, mode):
    """dX = (dZ W) act'(.), dZ heavy-tailed: rows log-uniform over 1e-8 .. 1 of the largest, 30 % of the rows exactly zero
    (clipped PPO samples) -- every row to 1e-5 of ITSELF (measured ~3e-7)"""
    from dtc_amd import h2i, ops
    g = torch.Generator().manual_seed(M + N + K)
    dZ = _rows(M, N, g, span=8, zero_frac=0.3)
    W = torch.randn(N, K, generator=g) / N ** 0.5
    Xs = torch.randn(M, K, generator=g)
    ref = dZ.double() @ W.double()
    kw = dict()
    if mode == "mask":
        mask = ops.relu_mask(M, K, DEV)
        Yf = torch.empty(M, K, device=DEV)
        # the sign record as the forward kernel writes it: a ReLU layer whose pre-activation is Xs
        eye_in = h2i.HImage.from_tensor(Xs.to(DEV))
        h2i.linear_fwd(eye_in, torch.eye(K, device=DEV), None, Yf, None, "relu", mask=mask)
        ref = ref * (Yf.double().cpu() > 0)
        kw = dict(mask=mask)
    elif mode == "elu":
        Ys = torch.nn.functional.elu(Xs)
        ref = torch.where(Ys.double() > 0, ref, ref * (Ys.double() + 1.0))
        kw = dict(Xsaved=Ys.to(DEV), act="elu")
    dX = torch.full((M, K), float("nan"), device=DEV)
    dXimg = h2i.HImage(M, K, DEV)
    h2i.linear_dgrad(h2i.HImage.from_tensor(dZ.to(DEV)), W.to(DEV), dX, dXimg, **kw)
    err = _row_err(dX, ref)
    print(f"dgrad {M}x{N}x{K} {mode}: per-row err {err:.2e}")
    assert err < 1e-5 and err < ROW_TOL
    assert float((dXimg.to_tensor() - dX).abs().max()) <= 2.0 ** -21 * float(dX.abs().max())


def test_dgrad_window_add_and_segmented_destination():
    """the actor's first layer backward: window [72, 584) -> d l_t as an image, with a second fp32 contribution added first; window
    [53, 72) -> the narrow blocks dz (16) | d mu (3, accumulating)"""
    from dtc_amd import h2i
    from dtc_amd._ffi import seg, segmat
    g = torch.Generator().manual_seed(3)
    M = 512
    dZ, W = torch.randn(M, 512, generator=g), torch.randn(512, 584, generator=g) / 22
    other = torch.randn(M, 512, generator=g)
    full = dZ.double() @ W.double()
    dZi, Wd = h2i.HImage.from_tensor(dZ.to(DEV)), W.to(DEV)
    dlt = h2i.HImage(M, 512, DEV)
    h2i.linear_dgrad(dZi, Wd, None, dlt, window=(72, 512), add=other.to(DEV))
    assert _row_err(dlt.to_tensor(), full[:, 72:] + other.double()) < ROW_TOL
    dz, dmu = torch.full((M, 16), float("nan"), device=DEV), torch.ones(M, 35, device=DEV)
    h2i.linear_dgrad(dZi, Wd, segmat([seg(dz, 0, 16), seg(dmu, 0, 3, accumulate=True)]), None, window=(53, 19))
    assert _row_err(dz, full[:, 53:69]) < ROW_TOL
    assert _row_err(dmu[:, :3], full[:, 69:72] + 1.0) < ROW_TOL and float((dmu[:, 3:] - 1).abs().max()) == 0.0
    # both windows in ONE launch: four image tiles + one fp32 tile
    dlt2 = h2i.HImage(M, 512, DEV)
    dz2, dmu2 = torch.full((M, 16), float("nan"), device=DEV), torch.zeros(M, 35, device=DEV)
    h2i.linear_dgrad(dZi, Wd, segmat([seg(None, 0, 512), seg(dz2, 0, 16), seg(dmu2, 0, 3)]), dlt2, window=[(72, 512), (53, 19)])
    assert _row_err(dlt2.to_tensor(), full[:, 72:]) < ROW_TOL and torch.equal(dz2, dz)
    assert _row_err(dmu2[:, :3], full[:, 69:72]) < ROW_TOL and float(dmu2[:, 3:].abs().max()) == 0.0


@pytest.mark.parametrize("M,shapes", [(1024, [(512, 512)]), (3072, [(512, 693), (693, 512), (128, 256)]), (24576, [(512, 512), (256, 512)]),
                                      (1000, [(140, 70)]),
                                      # 2 / 4 / 5 blocks of 128 rows per batch slice: the kernel's stage loop is unrolled over three blocks
                                      (2048, [(256, 140)]), (4096, [(128, 256), (12, 128)]), (5000, [(140, 256)])])
def test_wgrad_group_heavy_tailed(M, shapes):
    """dW = dZ^T X, db = colsum(dZ) with heavy-tailed dZ rows (1e-8 .. 1, 30 % zero) and X rows over 1e3: relative to every ROW of
    dW (one output feature) 2e-6, bias gradient likewise"""
    from dtc_amd import h2i, ops
    g = torch.Generator().manual_seed(M + len(shapes))
    jobs, refs = [], []
    for (N, K) in shapes:
        dZ = _rows(M, N, g, span=8, zero_frac=0.3)
        X = _rows(M, K, g, span=3)
        dW = torch.full((N, K + 8), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV)
        jobs.append((h2i.HImage.from_tensor(dZ.to(DEV)), h2i.HImage.from_tensor(X.to(DEV)), dW, 8, db))
        refs.append((dZ.double().T @ X.double(), dZ.double().sum(0)))
    ws = ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, M), DEV)
    h2i.wgrad_group(jobs, M, ws)
    for (dZi, Xi, dW, c0, db), (rW, rb) in zip(jobs, refs):
        eW = _row_err(dW[:, c0:], rW)
        eb = float(((db.double().cpu() - rb).abs() / rb.abs().clamp_min(1e-300)).max())
        scale_b = float((db.double().cpu() - rb).abs().max() / rb.abs().max())
        print(f"wgrad {M}x{dZi.K}x{Xi.K}: per-row err {eW:.2e}, bias {scale_b:.2e}")
        assert eW < ROW_TOL and scale_b < ROW_TOL
        assert bool(torch.isnan(dW[:, :c0]).all())                      # columns outside the job's window are untouched


def test_non_finite_elements_stay_in_their_rows_and_columns():
    from dtc_amd import h2i, ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 512, 256, 512
    X, W = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / 23
    Xb = X.clone()
    Xb[7, 100] = float("nan")
    Xb[300, 5] = float("inf")
    Xb[301] = float("nan")                                             # a whole row
    Wd = W.to(DEV)
    Y0, Y1 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    I0, I1 = h2i.HImage(M, N, DEV), h2i.HImage(M, N, DEV)
    h2i.linear_fwd(h2i.HImage.from_tensor(X.to(DEV)), Wd, None, Y0, I0, "elu")
    h2i.linear_fwd(h2i.HImage.from_tensor(Xb.to(DEV)), Wd, None, Y1, I1, "elu")
    bad = torch.zeros(M, dtype=torch.bool)
    bad[[7, 300, 301]] = True
    assert torch.equal(Y0[~bad.to(DEV)], Y1[~bad.to(DEV)])            # every other row: bit-identical
    assert bool((~torch.isfinite(Y1[bad.to(DEV)])).all())
    assert torch.equal(I0.to_tensor()[~bad.to(DEV)], I1.to_tensor()[~bad.to(DEV)])
    # through a second layer (the poisoned image as operand) and the data gradient
    W2 = (torch.randn(128, N, generator=g) / 16).to(DEV)
    Z0, Z1 = torch.empty(M, 128, device=DEV), torch.empty(M, 128, device=DEV)
    h2i.linear_fwd(I0, W2, None, Z0, None, None)
    h2i.linear_fwd(I1, W2, None, Z1, None, None)
    assert torch.equal(Z0[~bad.to(DEV)], Z1[~bad.to(DEV)]) and bool((~torch.isfinite(Z1[bad.to(DEV)])).all())
    D0, D1 = torch.empty(M, K, device=DEV), torch.empty(M, K, device=DEV)
    h2i.linear_dgrad(I0, Wd, D0)
    h2i.linear_dgrad(I1, Wd, D1)
    assert torch.equal(D0[~bad.to(DEV)], D1[~bad.to(DEV)]) and bool((~torch.isfinite(D1[bad.to(DEV)])).all())
    # weight gradient: a non-finite dZ[m, n] poisons row n of dW (all of it) and nothing else
    dZ = torch.randn(M, N, generator=g)
    dZb = dZ.clone()
    dZb[40, 17] = float("nan")
    Xi = h2i.HImage.from_tensor(X.to(DEV))
    outs = []
    for z in (dZ, dZb):
        dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        jobs = [(h2i.HImage.from_tensor(z.to(DEV)), Xi, dW, 0, db)]
        h2i.wgrad_group(jobs, M, ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, M), DEV))
        outs.append((dW, db))
    keep = torch.ones(N, dtype=torch.bool, device=DEV)
    keep[17] = False
    assert torch.equal(outs[0][0][keep], outs[1][0][keep]) and torch.equal(outs[0][1][keep], outs[1][1][keep])
    assert bool((~torch.isfinite(outs[1][0][17])).all()) and not bool(torch.isfinite(outs[1][1][17]))


def test_timing_next_to_the_converting_kernels(capsys):
    """informative: the image-operand kernels next to round 4's converting two-term kernels on the bench's largest layer"""
    from dtc_amd import h2i, ops
    M, N, K = 24576, 512, 512
    g = torch.Generator().manual_seed(0)
    X, W, b = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / 23).to(DEV), torch.randn(N, generator=g).to(DEV)
    Y = torch.empty(M, N, device=DEV)
    Xi, Yi, wset = h2i.HImage.from_tensor(X), h2i.HImage(M, N, DEV), h2i.WeightSet()
    dZi = h2i.HImage.from_tensor(torch.randn(M, N, generator=g).to(DEV))
    dXi = h2i.HImage(M, K, DEV)
    mask = ops.relu_mask(M, N, DEV)
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    jobs = [(dZi, Xi, dW, 0, db)] * 3
    wws = ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, M), DEV)

    def timed(fn, n=30):
        for _ in range(5):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    t = dict(
        fwd_img_to_img=timed(lambda: h2i.linear_fwd(Xi, W, b, None, Yi, "relu", mask=mask, wset=wset)),
        fwd_img_to_fp32=timed(lambda: h2i.linear_fwd(Xi, W, b, Y, None, "relu", wset=wset)),
        dgrad_img_to_img=timed(lambda: h2i.linear_dgrad(dZi, W, None, dXi, mask=mask, wset=wset)),
        wgrad_3x512x512=timed(lambda: h2i.wgrad_group(jobs, M, wws)),
        pack_512=timed(lambda: Xi.pack(X)),
        fwd_h2_converting=timed(lambda: ops.linear_fwd(X, W, b, Y, "relu", split=True)),
    )
    with capsys.disabled():
        print("\n[h2i timing, us] " + ", ".join(f"{k} {v:.1f}" for k, v in t.items()))


def test_images_written_by_the_latent_and_loss_kernels_equal_a_pack_of_their_fp32_outputs():
    """[z | mu[:, :3]] (dtc_cenet_latent_fwd_img) and dL/d recons (dtc_vae_loss_fused_img) leave their kernels as operand images
    too: byte for byte what h2i_pack_kernel makes of the fp32 tensors the same launches write (rows with a zero / huge entry incl.)."""
    from dtc_amd import _ffi, h2i, ops
    B = 1024 + 128
    g = torch.Generator().manual_seed(17)
    mulv = torch.randn(B, 35, generator=g)
    mulv[:, 19:] = 0.3 * mulv[:, 19:] - 1.0
    mulv[5, 19:] = 9.0                                   # outliers -> replaced by the median
    mulv[7, :3] = 0.0
    mulv[9, 0] = 3.0e4
    eps = torch.randn(B, 16, generator=g)
    eps[11] = 0.0
    mulv_d, eps_d = mulv.to(DEV), eps.to(DEV)
    z = torch.empty(B, 16, device=DEV)
    mask = torch.empty(B, 16, dtype=torch.uint8, device=DEV)
    info = torch.zeros(4, dtype=torch.int32, device=DEV)
    ws = torch.empty(int(_ffi.lib().dtc_cenet_workspace(B)) // 8 + 1, dtype=torch.float64, device=DEV)
    img = h2i.HImage(B, 19, DEV)
    ops.cenet_latent_fwd(mulv_d, eps_d, z, mask, info, ws, zmu_img=img)
    want = h2i.HImage.from_tensor(torch.cat([z, mulv_d[:, :3]], dim=1))
    assert torch.equal(img.buf, want.buf)
    # loss kernel
    rec, nxt = torch.randn(B, 53, generator=g).to(DEV), torch.randn(2 * B, 53, generator=g).to(DEV)
    rec[3] = nxt[3]                                      # an all-zero gradient row
    bv = torch.randn(2 * B, 3, generator=g).to(DEV)
    idx = torch.arange(B, device=DEV)
    d_rec, dmulv = torch.empty(B, 53, device=DEV), torch.empty(B, 35, device=DEV)
    losses = torch.zeros(4, device=DEV)
    lws = ops.workspace(_ffi.lib().dtc_loss_workspace(B), DEV)
    gimg = h2i.HImage(B, 53, DEV)
    ops.vae_loss_fused(rec, mulv_d, nxt, bv, idx, d_rec, dmulv, None, 0, losses, lws, drec_img=gimg)
    assert torch.equal(gimg.buf, h2i.HImage.from_tensor(d_rec).buf)
    assert float(d_rec[3].abs().max()) == 0.0


def test_images_written_by_the_fused_heads_kernel_equal_a_pack_of_its_fp32_outputs():
    """dtc_ppo_heads_loss_img: dHa / dHc [B, 128] and dmean [B, 12] / dvalue [B, 1] as operand images, byte for byte a pack of the fp32
    tensors the same launch writes (ragged last row tile, rows whose advantage is zero -> all-zero gradient rows)."""
    from dtc_amd import _ffi, h2i, ops
    B, H, A = 1024 + 37, 128, 12
    g = torch.Generator().manual_seed(23)
    r = lambda *s: torch.randn(*s, generator=g).to(DEV)          # noqa: E731
    Ha, Hc = torch.nn.functional.elu(r(B, H)), torch.nn.functional.elu(r(B, H))
    Wa, ba, Wc, bc = r(A, H) / 11, r(A) * 0.1, r(1, H) / 11, r(1) * 0.1
    std = torch.rand(A, generator=g).to(DEV) + 0.5
    R = 2 * B
    actions, old_mu = r(R, A), r(R, A)
    old_sigma = torch.rand(R, A, generator=g).to(DEV) + 0.5
    old_logp, adv, ret, oldv = r(R), r(R), r(R), r(R)
    idx = torch.randperm(R, generator=g)[:B].to(DEV)
    adv[idx[:20]] = 0.0
    cfg = _ffi.DtcPpoCfg()
    cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.desired_kl, cfg.use_clipped_value_loss, cfg.adaptive_schedule = 0.2, 1.0, 0.003, 0.01, 1, 0
    mean, val, dmean, dval = (torch.empty(B, w, device=DEV) for w in (A, 1, A, 1))
    dHa, dHc = torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
    dstd, losses = torch.zeros(A, device=DEV), torch.zeros(4, device=DEV)
    lr = torch.full((1,), 1e-3, dtype=torch.float64, device=DEV)
    ws = ops.workspace(_ffi.lib().dtc_loss_workspace(B), DEV)
    imgs = (h2i.HImage(B, H, DEV), h2i.HImage(B, H, DEV), h2i.HImage(B, A, DEV), h2i.HImage(B, 1, DEV))
    ops.ppo_heads_loss(Ha, Hc, Wa, ba, Wc, bc, "elu", std, actions, old_logp, old_mu, old_sigma, adv, ret, oldv, idx, cfg, mean, val,
                       dmean, dval, dHa, dHc, dstd, losses, lr, ws, imgs=imgs)
    for im, t in zip(imgs, (dHa, dHc, dmean, dval)):
        assert torch.equal(im.buf, h2i.HImage.from_tensor(t).buf)
    # the fp32 copies of dHa / dHc are optional where their images are given (the trainers' image chain passes NULL): same images, same
    # everything else
    imgs2 = (h2i.HImage(B, H, DEV), h2i.HImage(B, H, DEV), h2i.HImage(B, A, DEV), h2i.HImage(B, 1, DEV))
    mean2, val2, dmean2, dval2 = (torch.empty(B, w, device=DEV) for w in (A, 1, A, 1))
    dstd2, losses2 = torch.zeros(A, device=DEV), torch.zeros(4, device=DEV)
    ops.ppo_heads_loss(Ha, Hc, Wa, ba, Wc, bc, "elu", std, actions, old_logp, old_mu, old_sigma, adv, ret, oldv, idx, cfg, mean2, val2,
                       dmean2, dval2, None, None, dstd2, losses2, lr, ws, imgs=imgs2)
    for a, b in zip(imgs + (mean, val, dmean, dval, dstd, losses), imgs2 + (mean2, val2, dmean2, dval2, dstd2, losses2)):
        assert torch.equal(a.buf, b.buf) if isinstance(a, h2i.HImage) else torch.equal(a, b)
    with pytest.raises(_ffi.DtcError):           # neither an fp32 destination nor an image for dHa
        ops.ppo_heads_loss(Ha, Hc, Wa, ba, Wc, bc, "elu", std, actions, old_logp, old_mu, old_sigma, adv, ret, oldv, idx, cfg, mean2, val2,
                           dmean2, dval2, None, dHc, dstd2, losses2, lr, ws, imgs=(None, imgs2[1], None, None))


def test_64_row_tiles_equal_128_row_tiles_bit_for_bit(tile_rows):
    """The tile height is scheduling only: forward (fp32 + image result, sign record) and data gradient (image result) of the same
    operands are bit-identical on 64-row and 128-row tiles (same per-row arithmetic, same exponents)."""
    if tile_rows != "rows64_default":
        pytest.skip("one run compares both settings")
    from dtc_amd import _ffi, h2i, ops
    g = torch.Generator().manual_seed(31)
    M, N, K = 640, 128, 265
    X, W, b = _rows(M, K, g, span=6, zero_frac=0.1).to(DEV), (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), torch.randn(N, generator=g).to(DEV)
    dZ = _rows(M, N, g, span=6, zero_frac=0.2).to(DEV)
    out = []
    for mx in (-1, 0):
        _ffi.lib().dtc_h2i_rows64_max(mx)
        Xi, dZi = h2i.HImage.from_tensor(X), h2i.HImage.from_tensor(dZ)
        Y, Yi = torch.zeros(M, N, device=DEV), h2i.HImage(M, N, DEV)
        mask = ops.relu_mask(M, N, DEV).zero_()
        h2i.linear_fwd(Xi, W, b, Y, Yi, "relu", mask=mask)
        dXi = h2i.HImage(M, K, DEV)
        h2i.linear_dgrad(dZi, W, None, dXi)
        torch.cuda.synchronize()
        out.append((Y.clone(), Yi.buf.clone(), mask.clone(), dXi.buf.clone()))
    for a, c in zip(*out):
        assert torch.equal(a.view(torch.int32) if a.dtype == torch.float32 else a.view(torch.int64) if a.dtype == torch.float64 else a,
                           c.view(torch.int32) if c.dtype == torch.float32 else c.view(torch.int64) if c.dtype == torch.float64 else c)


@pytest.mark.parametrize("M", [640, 24576])
def test_narrow_layer_chains_equal_the_per_layer_launches_bit_for_bit(M):
    """dtc_linear_fwd_chain_h2i / dtc_linear_dgrad_chain_h2i: the CE-net's narrow stacks as ONE launch per direction (the workgroup of a
    row tile runs layer after layer on it) -- every result (images incl. exponents, fp32 outputs, sign records) equals the per-layer
    launches bit for bit: encoder 265 -> 128 (ReLU, sign record) -> 64 -> 35, decoder [19 | 512] -> 64 -> 128 -> 53, and the two
    data-gradient chains back through them (sign records applied)."""
    from dtc_amd import h2i, ops
    g = torch.Generator().manual_seed(41)
    dev = DEV
    w = lambda n, k: (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    bias = lambda n: torch.randn(n, generator=g).to(dev)
    hist = h2i.HImage.from_tensor(_rows(M, 265, g, span=3).to(dev))
    zmu, lt = h2i.HImage.from_tensor(torch.randn(M, 19, generator=g).to(dev)), h2i.HImage.from_tensor(_rows(M, 512, g, span=2).to(dev))
    We, be = [w(128, 265), w(64, 128), w(35, 64)], [bias(128), bias(64), bias(35)]
    Wd, bd = [w(64, 531), w(128, 64), w(53, 128)], [bias(64), bias(128), bias(53)]

    def run(chain):
        out = []
        # encoder forward
        e1, e = h2i.HImage(M, 128, dev), h2i.HImage(M, 64, dev)
        m1 = ops.relu_mask(M, 128, dev).zero_()
        mulv = torch.zeros(M, 35, device=dev)
        enc = [dict(X=hist, W=We[0], b=be[0], Yimg=e1, act="relu", mask=m1), dict(X=e1, W=We[1], b=be[1], Yimg=e),
               dict(X=e, W=We[2], b=be[2], Y=mulv)]
        # decoder forward (two-image input)
        c1, c2 = h2i.HImage(M, 64, dev), h2i.HImage(M, 128, dev)
        mc1, mc2 = ops.relu_mask(M, 64, dev).zero_(), ops.relu_mask(M, 128, dev).zero_()
        rec = torch.zeros(M, 53, device=dev)
        dec = [dict(X=[zmu, lt], W=Wd[0], b=bd[0], Yimg=c1, act="relu", mask=mc1), dict(X=c1, W=Wd[1], b=bd[1], Yimg=c2, act="relu", mask=mc2),
               dict(X=c2, W=Wd[2], b=bd[2], Y=rec)]
        if chain:
            h2i.linear_fwd_chain(enc)
            h2i.linear_fwd_chain(dec)
        else:
            for L in enc + dec:
                h2i.linear_fwd(L["X"], L["W"], L.get("b"), L.get("Y"), L.get("Yimg"), L.get("act"), L.get("mask"))
        out += [e1.buf, e.buf, m1, mulv, c1.buf, c2.buf, mc1, mc2, rec]
        # data-gradient chains: d mulv -> head^T -> ce1^T (sign record of e1); d recons -> cd2^T (c2's record) -> cd1^T (c1's record)
        dm = h2i.HImage.from_tensor(_rows(M, 35, torch.Generator().manual_seed(5), span=5, zero_frac=0.2).to(dev))
        dr = h2i.HImage.from_tensor(_rows(M, 53, torch.Generator().manual_seed(6), span=5).to(dev))
        g_head, g_ce1 = h2i.HImage(M, 64, dev), h2i.HImage(M, 128, dev)
        g_cd2, g_cd1 = h2i.HImage(M, 128, dev), h2i.HImage(M, 64, dev)
        benc = [dict(dZimg=dm, W=We[2], dXimg=g_head), dict(dZimg=g_head, W=We[1], dXimg=g_ce1, mask=m1)]
        bdec = [dict(dZimg=dr, W=Wd[2], dXimg=g_cd2, mask=mc2), dict(dZimg=g_cd2, W=Wd[1], dXimg=g_cd1, mask=mc1)]
        if chain:
            h2i.linear_dgrad_chain(benc)
            h2i.linear_dgrad_chain(bdec)
        else:
            for L in benc + bdec:
                h2i.linear_dgrad(L["dZimg"], L["W"], None, L["dXimg"], mask=L.get("mask"))
        torch.cuda.synchronize()
        out += [g_head.buf, g_ce1.buf, g_cd2.buf, g_cd1.buf]
        return [t.clone() for t in out]

    a, b = run(True), run(False)
    for i, (x, y) in enumerate(zip(a, b)):
        bits = lambda t: t.view(torch.int64) if t.dtype == torch.float64 else t.view(torch.int32) if t.dtype == torch.float32 else t
        assert torch.equal(bits(x), bits(y)), i
    assert float(a[3].abs().max()) > 0 and float(a[8].abs().max()) > 0


@pytest.mark.parametrize("M", [24576, 384, 200])
def test_wide_layer_chains_equal_the_per_layer_launches_bit_for_bit(M):
    """Round 6: a chain layer may be several column tiles wide -- the actor's / critic's tails (actor_critic_decoder.py:323-349) as ONE launch
    per direction: forward 512 -> 256 -> 128 (ELU, fp32 + image outputs), backward 128 -> 256 -> 512 (ELU derivative through the saved fp32
    activations).  Every output equals the per-layer launches bit for bit."""
    from dtc_amd import h2i
    g = torch.Generator().manual_seed(43)
    dev = DEV
    w = lambda n, k: (torch.randn(n, k, generator=g) / k ** 0.5).to(dev)
    bias = lambda n: torch.randn(n, generator=g).to(dev)
    x0 = h2i.HImage.from_tensor(_rows(M, 512, g, span=3).to(dev))
    W1, b1, W2, b2 = w(256, 512), bias(256), w(128, 256), bias(128)
    G3 = h2i.HImage.from_tensor(_rows(M, 128, torch.Generator().manual_seed(7), span=5, zero_frac=0.1).to(dev))

    def run(chain):
        o1, o2 = torch.zeros(M, 256, device=dev), torch.zeros(M, 128, device=dev)
        i1, i2 = h2i.HImage(M, 256, dev), h2i.HImage(M, 128, dev)
        fwd = [dict(X=x0, W=W1, b=b1, Y=o1, Yimg=i1, act="elu"), dict(X=i1, W=W2, b=b2, Y=o2, Yimg=i2, act="elu")]
        g2, g1 = h2i.HImage(M, 256, dev), h2i.HImage(M, 512, dev)
        x1 = torch.nn.functional.elu(torch.randn(M, 512, generator=torch.Generator().manual_seed(9))).to(dev)     # saved activation of the 512-wide layer
        if chain:
            h2i.linear_fwd_chain(fwd)
        else:
            for L in fwd:
                h2i.linear_fwd(L["X"], L["W"], L["b"], L["Y"], L["Yimg"], L["act"])
        bwd = [dict(dZimg=G3, W=W2, dXimg=g2, Xsaved=o1, act="elu"), dict(dZimg=g2, W=W1, dXimg=g1, Xsaved=x1, act="elu")]
        if chain:
            h2i.linear_dgrad_chain(bwd)
        else:
            for L in bwd:
                h2i.linear_dgrad(L["dZimg"], L["W"], None, L["dXimg"], Xsaved=L["Xsaved"], act=L["act"])
        torch.cuda.synchronize()
        return [t.clone() for t in (o1, o2, i1.buf, i2.buf, g2.buf, g1.buf)]

    a, b = run(True), run(False)
    for i, (x, y) in enumerate(zip(a, b)):
        bits = lambda t: t.view(torch.int64) if t.dtype == torch.float64 else t.view(torch.int32)
        assert torch.equal(bits(x), bits(y)), i
    assert float(a[1].abs().max()) > 0 and float(h2i.HImage(M, 512, dev).buf.abs().max()) >= 0
