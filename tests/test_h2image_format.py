"""The operand-image format (include/dtc_hip.h, csrc/h2i_core.hpp) against its numpy restatement oracle/h2image.py.
CPU: the restatement round-trips to 2^-21 of each row block's largest element and keeps inf / NaN in place.
GPU: dtc_h2i_pack and the image an image-writing GEMM epilogue produces are BYTE-identical to the restatement's encoding."""
import numpy as np
import pytest
import torch

from oracle import h2image as OH


def _cases():
    g = np.random.default_rng(3)
    for M, K in ((1, 1), (128, 16), (130, 17), (300, 693), (384, 512)):
        A = (g.standard_normal((M, K)) * 10.0 ** g.integers(-12, 5, size=(M, 1))).astype(np.float32)
        A[g.random((M, K)) < 0.1] = 0.0
        A[g.random(M) < 0.1] = 0.0
        yield M, K, A


def test_restatement_round_trips_per_row_block():
    for M, K, A in _cases():
        ch, ex = OH.encode(A)
        assert ch.shape == (-(-M // 128), -(-K // 16), 2, 256, 8) and ex.shape == (ch.shape[0], -(-ch.shape[1] // 8), 128)
        dec = OH.decode(ch, ex, M, K)
        P = np.zeros((M, ex.shape[1] * 128), dtype=np.float32)
        P[:, :K] = np.abs(A)
        blk = np.repeat(P.reshape(M, -1, 128).max(axis=2), 128, axis=1)[:, :K]
        assert np.all(np.abs(dec.astype(np.float64) - A) <= blk * 2.0 ** -21)
        full = OH.decode(ch, ex, ch.shape[0] * 128, ch.shape[1] * 16)
        assert not full[M:].any() and not full[:, K:].any()                     # padding rows / columns are zero
    # non-finite elements: the exponent comes from the finite ones, inf / NaN stay where they are
    A = np.ones((128, 128), dtype=np.float32)
    A[3, 5], A[3, 6], A[9, :] = np.inf, np.nan, np.nan
    ch, ex = OH.encode(A)
    assert ex[0, 0, 3] == 14 and ex[0, 0, 9] == OH.EZERO and ex[0, 0, 0] == 14
    dec = OH.decode(ch, ex, 128, 128)
    assert not np.isfinite(dec[3, 5]) and np.isnan(dec[3, 6]) and np.isnan(dec[9]).all() and np.array_equal(dec[0], A[0])


@pytest.mark.gpu
def test_kernels_write_exactly_the_restated_bytes():
    from dtc_amd import h2i
    dev = "cuda:0"

    def split_buf(img):
        rt, st = -(-img.M // 128), -(-img.K // 16)
        n = rt * st * 8192
        raw = img.buf.view(torch.uint8)
        return (raw[:n].cpu().numpy().view(np.uint16).reshape(rt, st, 2, 256, 8),
                raw[n:n + rt * (-(-st // 8)) * 512].cpu().numpy().view(np.int32).reshape(rt, -1, 128))

    for M, K, A in _cases():
        got_c, got_e = split_buf(h2i.HImage.from_tensor(torch.from_numpy(A).to(dev)))
        want_c, want_e = OH.encode(A)
        np.testing.assert_array_equal(got_e, want_e, err_msg=f"exponents {M} x {K}")
        np.testing.assert_array_equal(got_c, want_c, err_msg=f"dtc_h2i_pack {M} x {K}")
    # an image-writing epilogue: Y = elu(X W^T + b) written as fp32 AND as image; the image is the encoding of the fp32 result
    g = torch.Generator().manual_seed(4)
    X, W, b = torch.randn(300, 265, generator=g), torch.randn(140, 265, generator=g) / 16.0, torch.randn(140, generator=g)
    Y, Yimg = torch.empty(300, 140, device=dev), h2i.HImage(300, 140, dev)
    h2i.linear_fwd(h2i.HImage.from_tensor(X.to(dev)), W.to(dev), b.to(dev), Y, Yimg, "elu")
    want_c, want_e = OH.encode(Y.cpu().numpy())
    got_c, got_e = split_buf(Yimg)
    np.testing.assert_array_equal(got_e, want_e)
    np.testing.assert_array_equal(got_c, want_c)
