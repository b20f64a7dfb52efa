"""The activation-image format (include/dtc_hip.h) against its numpy restatement oracle/aimage.py.
CPU: the restatement round-trips, splits exactly as documented, and `ops.AImage.to_tensor` decodes its bytes.
GPU: `dtc_s3_aimage` and the image an image-writing GEMM epilogue produces are BYTE-identical to the restatement's encoding."""
import numpy as np
import pytest
import torch

from oracle import aimage as OA


def _cases():
    g = np.random.default_rng(3)
    for M, K in ((1, 1), (128, 16), (130, 17), (300, 693), (384, 512)):
        A = (g.standard_normal((M, K)) * 10.0 ** g.integers(-8, 5, size=(M, 1))).astype(np.float32)
        A[g.random((M, K)) < 0.1] = 0.0
        yield M, K, A


def test_restatement_round_trips_and_splits_exactly():
    for M, K, A in _cases():
        a1, a2, a3 = OA.split3(A)
        for pl in (a1, a2, a3):
            assert not np.any(pl.view(np.uint32) & 0xFFFF)                      # every term is a bf16 value
        s = (a1.astype(np.float64) + a2) + a3
        assert np.all(np.abs(s - A) <= np.abs(A) * 2.0 ** -24 + 1e-45)          # |a - (a1 + a2 + a3)| <= 2^-24 |a|
        img = OA.encode(A)
        assert img.shape == (-(-M // 128), -(-K // 16), 3, 256, 8)
        dec = OA.decode(img, M, K)
        assert np.all(np.abs(dec - A) <= np.abs(A) * 2.0 ** -23 + 1e-45)
        full = OA.decode(img, img.shape[0] * 128, img.shape[1] * 16)
        assert not full[M:].any() and not full[:, K:].any()                     # padding rows / columns are zero


def test_python_decoder_reads_the_same_bytes():
    from dtc_amd import ops
    for M, K, A in _cases():
        img = ops.AImage(M, K, "cpu")
        enc = OA.encode(A)
        img.buf.view(torch.int16).copy_(torch.from_numpy(enc.view(np.int16).reshape(-1)))
        np.testing.assert_array_equal(img.to_tensor().numpy(), OA.decode(enc, M, K))


@pytest.mark.gpu
def test_kernels_write_exactly_the_restated_bytes():
    from dtc_amd import ops
    dev = "cuda:0"
    for M, K, A in _cases():
        img = ops.AImage.from_tensor(torch.from_numpy(A).to(dev))
        got = img.buf.view(torch.int16).cpu().numpy().view(np.uint16).reshape(OA.encode(A).shape)
        np.testing.assert_array_equal(got, OA.encode(A), err_msg=f"dtc_s3_aimage {M} x {K}")
    # an image-writing epilogue: Y = X W^T + b written as fp32 AND as image; the image is the encoding of the fp32 result
    g = torch.Generator().manual_seed(4)
    X, W, b = torch.randn(300, 265, generator=g), torch.randn(140, 265, generator=g) / 16.0, torch.randn(140, generator=g)
    Y, Yimg = torch.empty(300, 140, device=dev), ops.AImage(300, 140, dev)
    ops.linear_fwd_img(ops.AImage.from_tensor(X.to(dev)), W.to(dev), b.to(dev), Y, Yimg, "elu")
    want = OA.encode(Y.cpu().numpy())
    got = Yimg.buf.view(torch.int16).cpu().numpy().view(np.uint16).reshape(want.shape)
    np.testing.assert_array_equal(got, want)
