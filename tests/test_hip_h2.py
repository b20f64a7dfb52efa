"""Two-term fp16 GEMM path (round 4; include/dtc_hip.h "two-term fp16 path", csrc/s3_core.hpp): the amax records its operands bring --
computed by the library, published by the producing kernels, static for unchanging tensors --, what the representation does with
operands of a wide dynamic range, and that a non-finite operand can never produce a silently wrong result.  The accuracy against fp64
next to the other two paths is in tests/test_hip_split.py (which runs on this path by default)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _fp16_terms():
    from dtc_amd import ops
    was = (ops.SPLIT, ops.H2)
    ops.set_split(True, h2=True)
    yield
    ops.set_split(was[0], h2=was[1])


def _bits(x):
    return int(torch.tensor(float(x), dtype=torch.float32).view(torch.int32).item()) & 0xffffffff


def _record(slot_ptr_or_tensor):
    """Largest word of an amax record (16 words, one per 128-byte line)."""
    t = slot_ptr_or_tensor
    return max(int(v) & 0xffffffff for v in t.view(-1).tolist())


def _amax(X, M, idx=None):
    from dtc_amd import _ffi, ops
    rec = torch.full((int(_ffi.lib().dtc_amax_record_bytes()) // 4,), 0x55, dtype=torch.int32, device=DEV)       # dtc_amax zeroes it itself
    _ffi.check(_ffi.lib().dtc_amax(ops.as_segmat(X, idx) if not isinstance(X, _ffi.DtcSegMat) else X, M, rec.data_ptr(), _ffi.stream()), "dtc_amax")
    torch.cuda.synchronize()
    return _record(rec)


def test_amax_is_the_exact_bit_pattern_of_the_largest_magnitude():
    from dtc_amd import _ffi
    g = torch.Generator().manual_seed(1)
    X = (torch.randn(3000, 133, generator=g) * 10.0 ** torch.randint(-8, 6, (3000, 1), generator=g).float()).to(DEV)
    assert _amax(X, 3000) == _bits(X.abs().max())
    assert _amax(X, 777) == _bits(X[:777].abs().max())                                   # only the rows below M
    V = X[:, 5:70]                                                                        # a column block of a wider matrix (row stride 133)
    assert _amax(_ffi.segmat([_ffi.seg(X, 5, 65)]), 3000) == _bits(V.abs().max())
    idx = torch.randint(0, 3000, (500,), generator=g).to(DEV)
    Xs = _ffi.segmat([_ffi.seg(X, 0, 133, gather=True), _ffi.seg(X[:500].contiguous(), 10, 7)], idx)
    want = max(float(X[idx].abs().max()), float(X[:500, 10:17].abs().max()))
    assert _amax(Xs, 500) == _bits(want)                                                  # gathered rows + a second segment
    assert _amax(torch.zeros(64, 16, device=DEV), 64) == 0
    Xn = X.clone()
    Xn[17, 3] = float("nan")
    assert _amax(Xn, 3000) > 0x7f800000                                                   # a NaN pattern outranks every number


@pytest.mark.parametrize("M,N,K", [(1024, 512, 512), (300, 256, 140), (24576, 512, 693)])
def test_published_amax_equals_the_tensors_amax_and_changes_no_bit(M, N, K):
    """Inside a WeightImages block a producing kernel adds the largest |value| it writes to the tensor's record and the consumers read
    it; outside a block the library computes the same number in front of the consumer: same exponent, same bits."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + K)
    X = torch.randn(M, K, generator=g).to(DEV)
    W1, W2 = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV), (torch.randn(256, N, generator=g) / N ** 0.5).to(DEV)
    b1 = torch.randn(N, generator=g).to(DEV)
    dZ = (torch.randn(M, 256, generator=g) * 1e-4).to(DEV)

    def run():
        Y, Z = torch.full((M, N), float("nan"), device=DEV), torch.full((M, 256), float("nan"), device=DEV)
        ops.linear_fwd(X, W1, b1, Y, "elu", split=True)
        ops.linear_fwd(Y, W2, None, Z, None, split=True)
        dY = torch.ones(M, N, device=DEV)
        ops.linear_dgrad(dZ, W2, ops.segmat([ops.seg(dY, 0, N, accumulate=True)]), split=True)       # sum of the old content and the product
        dX = torch.full((M, K), float("nan"), device=DEV)
        ops.linear_dgrad(dY, W1, dX, split=True)
        dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
        ws = ops.workspace(ops.wgrad_group_workspace_bytes([(dY, X, dW, db)], M, split=True), DEV)
        ops.wgrad_group([(dY, X, dW, db)], M, ws, split=True)
        return Y, Z, dY, dX, dW, db

    ref = run()
    imgs = ops.WeightImages()
    for _ in range(2):
        with imgs:
            out = run()
            reg = ops.amax_registry(torch.device(DEV))
            torch.cuda.synchronize()
            for t in (out[0], out[1], out[2], out[3]):                                    # Y, Z, dY (accumulated), dX: all published
                p = reg.of(t)
                assert p is not None
                w = (p - reg.arena.data_ptr()) // 4
                assert _record(reg.arena[w:w + reg.rec // 4]) == _bits(t.abs().max())
            assert reg.of(X) is None                                                      # nobody published X: the library computes it
        for a, b in zip(ref, out):
            assert torch.equal(a, b)


def test_static_amax_of_a_gathered_source_is_an_upper_bound_that_costs_nothing_measurable():
    """Trainers register the rollout storage once per update: the amax of ALL its rows stands in for the amax of a mini-batch's rows."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(3)
    R, B, K, N = 4000, 512, 693, 512
    S = torch.randn(R, K, generator=g)
    S[R - 1] *= 37.0                                                                      # the largest row is not in the mini-batch
    S = S.to(DEV)
    idx = torch.randint(0, R - 1, (B,), generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    ref = S[idx].double() @ W.double().T
    Y0, Y1 = torch.empty(B, N, device=DEV), torch.empty(B, N, device=DEV)
    ops.linear_fwd(_ffi.segmat([_ffi.seg(S, 0, K, gather=True)], idx), W, None, Y0, None, M=B, split=True)
    imgs = ops.WeightImages()
    try:
        ops.amax_static(S)
        with imgs:
            assert ops.amax_registry(torch.device(DEV)).of(S) is not None
            ops.linear_fwd(_ffi.segmat([_ffi.seg(S, 0, K, gather=True)], idx), W, None, Y1, None, M=B, split=True)
    finally:
        ops.amax_static_clear()
    err = lambda y: float((y.double() - ref).abs().max() / ref.abs().max())
    assert err(Y0) < 1.5e-6 and err(Y1) < 1.5e-6, (err(Y0), err(Y1))


def test_wide_dynamic_range_keeps_every_row_accurate_and_degrades_as_documented():
    """Rows whose magnitudes span 10^6: every row stays accurate RELATIVE TO ITSELF (the representation carries 22 significant bits for
    elements within 2^18 of the tensor's amax and an absolute error of 2^-40 amax below that -- include/dtc_hip.h)."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(5)
    M, N, K = 1024, 256, 512
    scale = 10.0 ** torch.linspace(-3, 3, M).unsqueeze(1)
    X = (torch.randn(M, K, generator=g) * scale).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    Y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(X, W, None, Y, None, split=True)
    ref = X.double() @ W.double().T
    row_err = ((Y.double() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1)).cpu()
    amax = float(X.abs().max())
    # a row r times below amax: relative error at most the fp32-chain level while r < 2^18, then growing as r 2^-40 (x sqrt(K) x slack)
    bound = torch.maximum(torch.full((M,), 2e-6), (amax / (scale.squeeze(1) * 3.0)) * 2.0 ** -40 * 64.0)
    assert bool((row_err <= bound).all()), (float(row_err.max()), int((row_err > bound).sum()))
    assert float(row_err[scale.squeeze(1) >= 1e-2].max()) < 2e-6                           # 10^5 below the largest rows: still the fp32 level


def test_non_finite_operand_turns_the_whole_result_nan():
    """fp32 kernels confine an inf / NaN to its rows and columns; here it also sits in the scale, so the WHOLE result is NaN: loud,
    never a silently wrong exponent."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(7)
    M, N, K = 384, 256, 256
    W = (torch.randn(N, K, generator=g) / 16.0).to(DEV)
    for bad in (float("inf"), float("nan")):
        X = torch.randn(M, K, generator=g).to(DEV)
        X[5, 9] = bad
        Y = torch.zeros(M, N, device=DEV)
        ops.linear_fwd(X, W, None, Y, None, split=True)
        assert bool(torch.isnan(Y).all())
        dX = torch.zeros(M, K, device=DEV)
        ops.linear_dgrad(torch.randn(M, N, generator=g).to(DEV), torch.where(torch.arange(K, device=DEV) == 3, torch.full_like(W, bad), W), dX, split=True)
        assert bool(torch.isnan(dX).all())
        dW, db = torch.zeros(N, K, device=DEV), torch.zeros(N, device=DEV)
        dZ = torch.randn(M, N, generator=g).to(DEV)
        ws = ops.workspace(ops.wgrad_group_workspace_bytes([(dZ, X, dW, db)], M, split=True), DEV)
        ops.wgrad_group([(dZ, X, dW, db)], M, ws, split=True)
        assert bool(torch.isnan(dW).all())


def test_all_zero_and_tiny_operands():
    from dtc_amd import ops
    M, N, K = 256, 128, 128
    W = (torch.randn(N, K) / 11.0).to(DEV)
    Y = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(torch.zeros(M, K, device=DEV), W, None, Y, None, split=True)
    assert bool((Y == 0).all())
    X = (torch.randn(M, K) * 1e-30).to(DEV)                                               # products near the bottom of fp32's range
    ops.linear_fwd(X, W, None, Y, None, split=True)
    ref = X.double() @ W.double().T
    assert float((Y.double() - ref).abs().max() / ref.abs().max()) < 2e-6
    Xd = torch.full((M, K), 1e-41, device=DEV)                                            # subnormal operand: amax has a zero exponent field
    ops.linear_fwd(Xd, torch.ones(N, K, device=DEV), None, Y, None, split=True)
    np.testing.assert_allclose(Y.cpu().numpy(), np.full((M, N), K * 1e-41, dtype=np.float32), rtol=1e-3)


def test_both_representations_stay_selectable_at_run_time():
    from dtc_amd import ops
    g = torch.Generator().manual_seed(9)
    M, N, K = 512, 512, 512
    X, W = torch.randn(M, K, generator=g).to(DEV), (torch.randn(N, K, generator=g) / 22.0).to(DEV)
    ref = X.double() @ W.double().T
    out = {}
    for h2 in (True, False):
        ops.set_split(True, h2=h2)
        Y = torch.empty(M, N, device=DEV)
        ops.linear_fwd(X, W, None, Y, None, split=True)
        out[h2] = Y
        assert float((Y.double() - ref).abs().max() / ref.abs().max()) < 1.5e-6
    assert not torch.equal(out[True], out[False])                                         # different sums, same error level
