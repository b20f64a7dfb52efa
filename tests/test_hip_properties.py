"""Property tests (hypothesis) of the HIP kernels against the oracle on randomly drawn shapes and edge patterns
(SURVEY.md §4.3): GAE with arbitrary T / N / done patterns / discount factors, row gather with ragged widths and
repeated indices, the planner's tie and sentinel rules, dense layers with random segmentations, the fused transition
store.  GPU only; every example goes through the C ABI."""
import numpy as np
import pytest
import torch
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

from dtc_amd import foothold, ops, synthetic as S
from dtc_amd._ffi import seg, segmat

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(deadline=None, max_examples=25, suppress_health_check=list(HealthCheck), derandomize=True)


@settings(**CFG)
@given(T=st.integers(1, 40), N=st.integers(1, 700), seed=st.integers(0, 2 ** 16), p_done=st.sampled_from([0.0, 0.02, 0.5, 1.0]),
       gamma=st.sampled_from([0.0, 0.9, 0.99, 1.0]), lam=st.sampled_from([0.0, 0.95, 1.0]))
def test_gae_scan_bit_exact_for_any_shape(T, N, seed, p_done, gamma, lam):
    from oracle import gae as OG
    g = torch.Generator().manual_seed(seed)
    rew, val = torch.randn(T, N, 1, generator=g), torch.randn(T, N, 1, generator=g)
    dones = (torch.rand(T, N, 1, generator=g) < p_done).to(torch.uint8)
    last = torch.randn(N, 1, generator=g)
    ref = OG.gae_scan(rew[..., 0].numpy(), val[..., 0].numpy(), dones[..., 0].numpy(), last[:, 0].numpy(), gamma, lam)
    ret, adv = torch.empty(T, N, 1, device=DEV), torch.empty(T, N, 1, device=DEV)
    stats = torch.zeros(4, dtype=torch.float64, device=DEV)
    ops.gae(rew.to(DEV), val.to(DEV), dones.to(DEV), last.to(DEV), gamma, lam, ret, adv, stats)
    np.testing.assert_array_equal(ret.cpu().numpy()[..., 0], ref)
    np.testing.assert_array_equal(adv.cpu().numpy()[..., 0], (ref - val[..., 0].numpy()).astype(np.float32))


@settings(**CFG)
@given(rows=st.integers(1, 3000), width=st.integers(1, 1500), n_idx=st.integers(0, 4000), seed=st.integers(0, 2 ** 16),
       dtype=st.sampled_from([torch.float32, torch.uint8, torch.int64]))
def test_gather_rows_any_width_any_index_multiset(rows, width, n_idx, seed, dtype):
    g = torch.Generator().manual_seed(seed)
    src = (torch.randn(rows, width, generator=g) * 100).to(dtype)
    idx = torch.randint(0, rows, (n_idx,), generator=g)            # repeats and gaps allowed
    out = ops.gather_rows(src.to(DEV), idx.to(DEV))
    assert torch.equal(out.cpu(), src[idx])


@settings(**{**CFG, "max_examples": 12})
@given(seed=st.integers(0, 2 ** 16), n=st.integers(1, 97),
       pattern=st.sampled_from(["flat", "all_exceptional", "far_outside", "checker", "one_valid_point", "random"]))
def test_planner_tie_and_sentinel_rules(seed, n, pattern):
    """Flat terrain -> many equal totals -> lowest flat index (torch.topk on CPU); every point exceptional -> index 0;
    nominal foothold outside the grid -> full scan among sentinels; all bit-exact against the oracle."""
    from oracle import foothold as OF
    inp = S.scorer_inputs(n, seed=seed)
    mh, root = inp["measured_heights"], inp["root_states"]
    if pattern == "flat":
        mh[:] = root[:, 2:3] - 0.32
    elif pattern == "all_exceptional":
        mh[:] = root[:, 2:3] + 2.0
    elif pattern == "far_outside":
        inp["thigh_pos"][:, :, :2] += 7.0
    elif pattern == "checker":
        mh[:, ::2] = root[:, 2:3] - 3.0
    elif pattern == "one_valid_point":
        mh[:] = root[:, 2:3] + 2.0
        mh[:, 346] = root[:, 2] - 0.32
    o = OF.plan(mh.numpy(), root.numpy(), inp["thigh_pos"].numpy(), inp["commands"].numpy(), S.MEASURED_POINTS_X,
                S.MEASURED_POINTS_Y)
    h = foothold.plan(mh.to(DEV), root.to(DEV), inp["thigh_pos"].to(DEV), inp["commands"].to(DEV))
    idx = h["optimal_foothold_indice"].squeeze(1).cpu().numpy()
    np.testing.assert_array_equal(idx, o["idx"])
    np.testing.assert_array_equal(h["foothold_obs"].cpu().numpy(), o["foothold_obs"])
    np.testing.assert_array_equal(h["optimal_footholds_world"].cpu().numpy(), o["optimal_footholds_world"])
    if pattern == "all_exceptional":
        assert (idx == 0).all()


@settings(**{**CFG, "max_examples": 20})
@given(M=st.integers(1, 700), N=st.integers(1, 200), widths=st.lists(st.integers(1, 90), min_size=1, max_size=4),
       gather=st.booleans(), act=st.sampled_from([None, "relu", "elu"]), seed=st.integers(0, 2 ** 16))
def test_dense_layer_any_segmentation(M, N, widths, gather, act, seed):
    """Y = act(cat(segments)[idx] W^T + b), its data gradient and weight gradient for random shapes / segmentations:
    every row / column / K tail, gathered or not, against torch (fp32 GEMM: 2e-5 of the result's scale)."""
    g = torch.Generator().manual_seed(seed)
    K = sum(widths)
    rows_src = M + 13
    srcs = [torch.randn(rows_src, w + 5, generator=g) for w in widths]          # segments live inside wider matrices
    idx = torch.randint(0, rows_src, (M,), generator=g) if gather else None
    W, b = torch.randn(N, K, generator=g) / K ** 0.5, torch.randn(N, generator=g)
    dZ = torch.randn(M, N, generator=g)
    rows = idx if gather else torch.arange(M)
    X = torch.cat([s[rows, 3:3 + w] for s, w in zip(srcs, widths)], dim=1)
    Yref = X @ W.t() + b
    Yref = torch.relu(Yref) if act == "relu" else (torch.nn.functional.elu(Yref) if act == "elu" else Yref)
    d = [s.to(DEV) for s in srcs]
    Xs = segmat([seg(t, 3, w, gather=gather) for t, w in zip(d, widths)], idx.to(DEV) if gather else None)
    Y = torch.empty(M, N, device=DEV)
    ops.linear_fwd(Xs, W.to(DEV), b.to(DEV), Y, act, M=M)
    tol = lambda ref: dict(rtol=2e-5, atol=2e-5 * max(1.0, float(ref.abs().max())))
    np.testing.assert_allclose(Y.cpu().numpy(), Yref.numpy(), **tol(Yref))
    dW, db = torch.empty(N, K, device=DEV), torch.empty(N, device=DEV)
    ws = ops.workspace(ops.wgrad_workspace_bytes(M, N, K), DEV)
    ops.linear_wgrad(dZ.to(DEV), Xs, dW, db, ws, M=M)
    np.testing.assert_allclose(dW.cpu().numpy(), (dZ.t() @ X).numpy(), **tol(dZ.t() @ X))
    np.testing.assert_allclose(db.cpu().numpy(), dZ.sum(0).numpy(), **tol(dZ.sum(0)))
    dX = torch.empty(M, K, device=DEV)
    ops.linear_dgrad(dZ.to(DEV), W.to(DEV), dX, None, None, M=M)
    np.testing.assert_allclose(dX.cpu().numpy(), (dZ @ W).numpy(), **tol(dZ @ W))
