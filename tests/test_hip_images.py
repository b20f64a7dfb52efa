"""Activation images (include/dtc_hip.h, round 4): the *_i3 kernels -- both operands read as bf16 x 3 LDS planes by LDS-DMA, results
written as fp32 and / or as the image the next kernel reads -- against fp64, next to the split kernels that convert inside their
K loop (csrc/gemm_s3.hip): same accuracy, and the image a kernel writes decodes to exactly the fp32 values it would have stored."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _bf16x3_split_kernels():
    """The activation-image chain is bf16 x 3: the converting kernels these tests put next to it run in the same representation
    (the default since round 4 is the two-term fp16 one, whose sums differ in the last bits)."""
    from dtc_amd import ops
    was = ops.H2
    ops.set_split(ops.SPLIT, h2=False)
    yield
    ops.set_split(ops.SPLIT, h2=was)


def _err(y, ref):
    return float((y.double().cpu() - ref).abs().max() / (ref.abs().max() + 1e-30))


def _act(v, act):
    return torch.relu(v) if act == "relu" else torch.nn.functional.elu(v) if act == "elu" else v


def _same_as_fp32(img, Y):
    """The image of a result decodes to the fp32 result: three bf16 terms carry 24 significant bits (one ulp of slack for values
    whose third remainder needs a ninth bit)."""
    got, want = img.to_tensor(), Y[:img.M, :img.K]
    assert got.shape == want.shape
    tol = want.abs() * 2.0 ** -23 + 1e-37
    bad = (got - want).abs() > tol
    assert not bool(bad.any()), (int(bad.sum()), float((got - want).abs().max()))


@pytest.mark.parametrize("M,K", [(128, 16), (300, 693), (24576, 512), (1, 5)])
def test_image_round_trip(M, K):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + K)
    X = (torch.randn(M, K + 3, generator=g) * 10.0 ** torch.randint(-6, 4, (M, 1), generator=g).float()).to(DEV)[:, :K]     # row stride != K
    img = ops.AImage.from_tensor(X)
    _same_as_fp32(img, X)
    # rows / columns past the matrix are zero inside the image's last chunks
    raw = img.buf.view(torch.int16)
    full = ops.AImage(-(-M // 128) * 128, -(-K // 16) * 16, DEV)
    full.buf.copy_(img.buf)
    dec = full.to_tensor()
    assert float(dec[M:].abs().max() if M % 128 else 0.0) == 0.0 and float(dec[:, K:].abs().max() if K % 16 else 0.0) == 0.0
    assert raw.numel() * 2 == full.buf.numel() * 8


@pytest.mark.parametrize("M,N,K,act", [(1024, 512, 512, "relu"), (384, 512, 693, "relu"), (300, 693, 512, None), (1000, 256, 512, "elu"),
                                       (24576, 512, 512, "elu"), (130, 128, 265, "relu"), (200, 140, 70, "elu")])
def test_forward_from_image_matches_the_split_kernel_and_writes_its_image(M, N, K, act):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + N + 3 * K)
    X = torch.randn(M, K, generator=g) * 10.0 ** torch.randint(-3, 3, (M, 1), generator=g).float()
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = _act(X.double() @ W.double().T + b.double(), act)
    Xd, Wd, bd = X.to(DEV), W.to(DEV), b.to(DEV)
    ys3 = torch.full((M, N), float("nan"), device=DEV)
    yi3 = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(Xd, Wd, bd, ys3, act, split=True)
    Yimg = ops.AImage(M, N, DEV)
    Yimg.buf.view(torch.int16).fill_(0x7FC0)                                   # bf16 NaN everywhere: every chunk must be written
    ops.linear_fwd_img(ops.AImage.from_tensor(Xd), Wd, bd, yi3, Yimg, act)
    es3, ei3 = _err(ys3, ref), _err(yi3, ref)
    print(f"fwd {M}x{N}x{K}: split err {es3:.2e}, image-operand err {ei3:.2e}")
    assert ei3 <= 2.0 * es3 + 2e-7
    _same_as_fp32(Yimg, yi3)
    pad = ops.AImage(-(-M // 128) * 128, -(-N // 16) * 16, DEV)
    pad.buf.copy_(Yimg.buf)
    dec = pad.to_tensor()
    assert not bool(torch.isnan(dec).any())
    assert (M % 128 == 0 or float(dec[M:].abs().max()) == 0.0) and (N % 16 == 0 or float(dec[:, N:].abs().max()) == 0.0)
    # image only / fp32 only give the same bits
    y2 = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd_img(ops.AImage.from_tensor(Xd), Wd, bd, y2, None, act)
    img2 = ops.AImage(M, N, DEV)
    ops.linear_fwd_img(ops.AImage.from_tensor(Xd), Wd, bd, None, img2, act)
    assert torch.equal(y2, yi3) and torch.equal(img2.buf.view(torch.int64), Yimg.buf.view(torch.int64))


def test_forward_from_image_sign_record_and_chain():
    """Two layers chained through an image only (the fp32 activation never exists), with the sign record of the first."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(9)
    B = 1536
    X, W1, b1, W2, b2 = (torch.randn(B, 512, generator=g), torch.randn(512, 512, generator=g) / 22.0, torch.randn(512, generator=g),
                         torch.randn(256, 512, generator=g) / 22.0, torch.randn(256, generator=g))
    h_ref = torch.relu(X.double() @ W1.double().T + b1.double())
    y_ref = h_ref @ W2.double().T + b2.double()
    d = lambda t: t.to(DEV)
    Himg, mask = ops.AImage(B, 512, DEV), ops.relu_mask(B, 512, DEV)
    ops.linear_fwd_img(ops.AImage.from_tensor(d(X)), d(W1), d(b1), None, Himg, "relu", mask=mask)
    y = torch.empty(B, 256, device=DEV)
    ops.linear_fwd_img(Himg, d(W2), d(b2), y, None, None)
    assert _err(y, y_ref) <= 3e-6
    h = Himg.to_tensor()
    pos = (h > 0).cpu().view(B // 32, 4, 2, 4, 512).permute(0, 2, 1, 3, 4).reshape(B // 32, 2, 16, 512).to(torch.int32)
    want = (pos << torch.arange(16, dtype=torch.int32).view(1, 1, 16, 1)).sum(dim=2).reshape(-1, 512)
    assert torch.equal(mask.cpu().view(-1, 512).to(torch.int32) & 0xFFFF, want)


@pytest.mark.parametrize("M,N,K", [(384, 693, 512), (1000, 140, 256)])
def test_mse_output_layer_from_image(M, N, K):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M)
    X, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    T = torch.randn(2 * M, N + 700, generator=g)
    idx = torch.randint(0, 2 * M, (M,), generator=g)
    e = (X.double() @ W.double().T + b.double()) - T[idx][:, 696:696 + N].double()
    d = lambda t: t.to(DEV)
    dY, dYimg = torch.full((M, N), float("nan"), device=DEV), ops.AImage(M, N, DEV)
    part = torch.zeros(int(ops.lib().dtc_linear_fwd_mse_s3_parts(M, N)), dtype=torch.float64, device=DEV)
    n = ops.linear_fwd_mse_img(ops.AImage.from_tensor(d(X)), d(W), d(b), d(T), 696, d(idx), dY, dYimg, part)
    assert _err(dY, e * (2.0 / (M * N))) <= 2e-6
    assert abs(float(part[:n].sum()) - float((e * e).sum())) <= 2e-6 * float((e * e).sum())
    _same_as_fp32(dYimg, dY)


@pytest.mark.parametrize("M,N,K,act,mode", [(1024, 512, 512, "relu", "mask"), (384, 693, 512, "relu", "mask"), (512, 256, 512, "elu", "saved"),
                                            (300, 128, 256, "elu", "saved"), (24576, 512, 512, "relu", "mask"), (640, 512, 512, None, "acc"),
                                            (200, 70, 140, None, "acc")])
def test_data_gradient_from_image(M, N, K, act, mode):
    """dX [M, K] = (dZ [M, N] W [N, K]) * act'(.) with dZ as an image; derivative through the sign record / the saved output;
    accumulation into an fp32 destination whose sum leaves as an image."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(2 * M + N + K)
    dZ = torch.randn(M, N, generator=g) * 10.0 ** torch.randint(-6, 0, (M, 1), generator=g).float()
    W = torch.randn(N, K, generator=g) / N ** 0.5
    Xs = _act(torch.randn(M, K, generator=g), act)
    ref = dZ.double() @ W.double()
    d = lambda t: t.to(DEV)
    dZimg = ops.AImage.from_tensor(d(dZ))
    dX, dXimg = torch.full((M, K), float("nan"), device=DEV), ops.AImage(M, K, DEV)
    if mode == "acc":
        old = torch.randn(M, K, generator=g) * 1e-3
        dst = d(old).clone()
        ops.linear_dgrad_img(dZimg, d(W), dst, dXimg, accumulate=True)
        assert torch.equal(dst.cpu(), old)                                     # only read: the sum exists as the image
        got = dXimg.to_tensor()
        assert _err(got, ref + old.double()) <= 2e-6
        dst2 = d(old).clone()
        ops.linear_dgrad_img(dZimg, d(W), dst2, None, accumulate=True)         # no image: the fp32 destination holds the sum
        _same_as_fp32(dXimg, dst2)
        return
    ref = ref * (Xs > 0) if act == "relu" else torch.where(Xs > 0, ref, ref * (Xs.double() + 1.0))
    ds3 = torch.full((M, K), float("nan"), device=DEV)
    if mode == "mask":
        Y = torch.empty(M, K, device=DEV)
        mask = ops.relu_mask(M, K, DEV) if (M % 128 == 0 and K % 128 == 0) else None
        if mask is None:
            pytest.skip("sign records need whole 128 x 128 tiles")
        # a sign record with the pattern of Xs: a forward whose output IS Xs (identity weights are not needed: write the record by hand)
        pos = (Xs > 0).view(M // 32, 4, 2, 4, K).permute(0, 2, 1, 3, 4).reshape(M // 32, 2, 16, K).to(torch.int32)
        words = (pos << torch.arange(16, dtype=torch.int32).view(1, 1, 16, 1)).sum(dim=2).reshape(-1)
        mask.copy_(torch.from_numpy(words.numpy().astype(np.uint16).view(np.int16)).to(DEV))
        ops.linear_dgrad(d(dZ), d(W), ds3, None, "relu", mask=mask, split=True)
        ops.linear_dgrad_img(dZimg, d(W), dX, dXimg, mask=mask)
    else:
        ops.linear_dgrad(d(dZ), d(W), ds3, d(Xs), act, split=True)
        ops.linear_dgrad_img(dZimg, d(W), dX, dXimg, Xsaved=d(Xs), act=act)
    es3, ei3 = _err(ds3, ref), _err(dX, ref)
    print(f"dgrad {M}x{N}x{K}: split err {es3:.2e}, image-operand err {ei3:.2e}")
    assert ei3 <= 2.0 * es3 + 2e-7
    _same_as_fp32(dXimg, dX)


@pytest.mark.parametrize("M,layers", [(1024, [(512, 512)]), (384, [(693, 512), (64, 128)]), (24576, [(512, 512), (256, 512)]),
                                      (1000, [(140, 70), (512, 693)]), (24576, [(512, 693), (512, 512), (512, 512)])])
def test_weight_gradients_from_images(M, layers):
    """dW = dZ^T X, db = colsum(dZ) for a bucket of layers, both operands as images, against fp64 next to the split kernel that
    converts inside its K loop."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + len(layers))
    jobs_s3, jobs_i3, refs = [], [], []
    for N, K in layers:
        dZ = (torch.randn(M, N, generator=g) * 10.0 ** torch.randint(-6, 0, (M, 1), generator=g).float()).to(DEV)
        X = torch.randn(M, K, generator=g).to(DEV)
        refs.append((dZ.double().T @ X.double(), dZ.double().sum(0)))
        mk = lambda: (torch.full((N, K), float("nan"), device=DEV), torch.full((N,), float("nan"), device=DEV))
        jobs_s3.append((dZ, X, *mk()))
        jobs_i3.append((ops.AImage.from_tensor(dZ), ops.AImage.from_tensor(X), *mk()))
    ws = ops.workspace(ops.wgrad_group_workspace_bytes(jobs_s3, M, split=True), DEV)
    ops.wgrad_group(jobs_s3, M, ws, split=True)
    wi = ops.workspace(ops.wgrad_group_img_workspace_bytes(jobs_i3, M), DEV)
    ops.wgrad_group_img(jobs_i3, M, wi)
    for (N, K), s3, i3, (rW, rb) in zip(layers, jobs_s3, jobs_i3, refs):
        eW3, eWi = _err(s3[2], rW.cpu()), _err(i3[2], rW.cpu())
        eb3, ebi = _err(s3[3], rb.cpu()), _err(i3[3], rb.cpu())
        print(f"wgrad {M}x{N}x{K}: dW split {eW3:.2e} image {eWi:.2e};  db split {eb3:.2e} image {ebi:.2e}")
        assert eWi <= 2.0 * eW3 + 2e-7 and ebi <= 2.0 * eb3 + 3e-7


@pytest.mark.parametrize("M,N,K,act", [(1024, 512, 693, "relu"), (384, 512, 584, "elu"), (300, 256, 140, None)])
def test_converting_forward_kernel_writes_images_too(M, N, K, act):
    """linear_s3_kernel (fp32 / gathered / segmented X, converted in its K loop) with an image result: the hand-over from the fp32 inputs
    of a stack (terrain heights, packed observations) into the image chain."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(M + K)
    R = 3 * M
    src, W, b = torch.randn(R, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), torch.randn(N, generator=g).to(DEV)
    idx = torch.randint(0, R, (M,), generator=g).to(DEV)
    X = _ffi.segmat([_ffi.seg(src, 0, K, gather=True)], idx)
    y0 = torch.empty(M, N, device=DEV)
    ops.linear_fwd(X, W, b, y0, act, M=M, split=True)
    y1, img, img2 = torch.full((M, N), float("nan"), device=DEV), ops.AImage(M, N, DEV), ops.AImage(M, N, DEV)
    mask = ops.relu_mask(M, N, DEV) if (act == "relu" and M % 128 == 0 and N % 128 == 0) else None
    ops.linear_fwd(X, W, b, y1, act, M=M, Yimg=img, mask=mask)
    ops.linear_fwd(X, W, b, None, act, M=M, Yimg=img2, mask=mask)
    assert torch.equal(y0, y1) and torch.equal(img.buf.view(torch.int64), img2.buf.view(torch.int64))
    _same_as_fp32(img, y1)


def test_converting_data_gradient_writes_the_image_of_one_destination_block():
    """The actor's first layer (ppo.py:333 backward through actor_body[0]): destination [None 53 | dz 16 | dmu 3 (accumulating) | dlt 512],
    with dlt leaving as an image for the terrain encoder's image-operand backward."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(8)
    B = 384
    dZ = torch.randn(B, 512, generator=g).to(DEV)
    Wa = (torch.randn(512, 584, generator=g) / 22.0).to(DEV)
    full = (dZ.double() @ Wa.double()).cpu()
    mk = lambda: (torch.zeros(B, 16, device=DEV), torch.full((B, 35), 0.5, device=DEV), torch.empty(B, 512, device=DEV))
    dz0, dmu0, dlt0 = mk()
    ops.linear_dgrad(dZ, Wa, _ffi.segmat([_ffi.seg(None, 0, 53), _ffi.seg(dz0, 0, 16), _ffi.seg(dmu0, 0, 3, accumulate=True), _ffi.seg(dlt0, 0, 512)]), split=True)
    dz1, dmu1, dlt1 = mk()
    img = ops.AImage(B, 512, DEV)
    ops.linear_dgrad(dZ, Wa, _ffi.segmat([_ffi.seg(None, 0, 53), _ffi.seg(dz1, 0, 16), _ffi.seg(dmu1, 0, 3, accumulate=True), _ffi.seg(dlt1, 0, 512)]),
                     dXimg=img, img_seg=3)
    assert torch.equal(dz0, dz1) and torch.equal(dmu0, dmu1) and torch.equal(dlt0, dlt1)
    np.testing.assert_allclose(dlt1.cpu().numpy(), full[:, 72:].numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(dmu1.cpu().numpy()[:, :3], 0.5 + full[:, 69:72].numpy(), rtol=2e-5, atol=2e-5)
    _same_as_fp32(img, dlt1)
    # accumulating image block: fp32 only read, the sum in the image; single block with the ELU derivative
    old = torch.randn(B, 512, generator=g).to(DEV)
    dst, img2 = old.clone(), ops.AImage(B, 512, DEV)
    Wt = (torch.randn(512, 512, generator=g) / 22.0).to(DEV)
    ops.linear_dgrad(dZ, Wt, _ffi.segmat([_ffi.seg(dst, 0, 512, accumulate=True)]), dXimg=img2, img_seg=0)
    assert torch.equal(dst, old)
    want = old.clone()
    ops.linear_dgrad(dZ, Wt, _ffi.segmat([_ffi.seg(want, 0, 512, accumulate=True)]), split=True)
    _same_as_fp32(img2, want)
    Xs = torch.nn.functional.elu(torch.randn(B, 512, generator=g)).to(DEV)
    d0, d1, img3 = torch.empty(B, 512, device=DEV), torch.empty(B, 512, device=DEV), ops.AImage(B, 512, DEV)
    ops.linear_dgrad(dZ, Wt, d0, Xs, "elu", split=True)
    ops.linear_dgrad(dZ, Wt, d1, Xs, "elu", dXimg=img3)
    assert torch.equal(d0, d1)
    _same_as_fp32(img3, d1)
