"""HIP kernels (through the C ABI) vs the CPU oracle on the same seeded inputs.  GPU only."""
import numpy as np
import pytest
import torch

from dtc_amd import synthetic as S

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _np(t):
    return t.detach().cpu().numpy()


# ------------------------------------------------------------------------------ scorer
def _scorer_case(inp, debug=True):
    from dtc_amd import foothold
    from oracle import foothold as OF
    o = OF.plan(_np(inp["measured_heights"]), _np(inp["root_states"]), _np(inp["thigh_pos"]), _np(inp["commands"]),
                S.MEASURED_POINTS_X, S.MEASURED_POINTS_Y, want_debug=debug)
    d = {k: v.to(DEV) for k, v in inp.items()}
    h = foothold.plan(d["measured_heights"], d["root_states"], d["thigh_pos"], d["commands"], want_debug=debug)
    torch.cuda.synchronize()
    return o, h


def _assert_scorer_equal(o, h, debug=True):
    np.testing.assert_array_equal(_np(h["optimal_foothold_indice"]).squeeze(1), o["idx"])
    np.testing.assert_array_equal(_np(h["foothold_obs"]), o["foothold_obs"])
    np.testing.assert_array_equal(_np(h["optimal_footholds_world"]), o["optimal_footholds_world"])
    np.testing.assert_array_equal(_np(h["pred_footholds"]), o["pred_footholds"])
    np.testing.assert_array_equal(_np(h["pred_footholds_to_robot"]), o["pred_footholds_to_robot"])
    if debug:
        np.testing.assert_array_equal(_np(h["foothold_score"]), o["foothold_score"])
        np.testing.assert_array_equal(_np(h["slope"]), o["slope"])
        np.testing.assert_array_equal(_np(h["nominal_footholds_indice"]), o["nominal_idx"])
        np.testing.assert_array_equal(_np(h["heights_world"])[:, :, :2], o["heights_world_xy"])


@pytest.mark.parametrize("N", [1, 3, 4, 5, 257, 4096])
def test_scorer_bit_exact_vs_oracle(N):
    o, h = _scorer_case(S.scorer_inputs(N, seed=7 + N))
    _assert_scorer_equal(o, h)


@pytest.mark.parametrize("N", [1, 2, 3, 4, 5, 7, 257, 4096, 5003])
def test_scorer_fast_kernel_bit_exact_vs_oracle(N):
    """Without debug outputs the 33 x 21 grid takes the persistent fast kernel (wave-owned env ranges, groups of 4)."""
    o, h = _scorer_case(S.scorer_inputs(N, seed=11 + N), debug=False)
    _assert_scorer_equal(o, h, debug=False)


def _stress_inputs(kind, N=3072, seed=21):
    inp = S.scorer_inputs(N, seed=seed)
    g = torch.Generator().manual_seed(seed + 1)
    if kind == "border":          # fast base velocities / commands push the nominal footholds to and past the grid border
        inp["root_states"][:, 7:10] *= 12.0
        inp["commands"][:, :2] *= 8.0
        inp["thigh_pos"][:, :, :2] += 0.5 * (torch.rand(N, 4, 2, generator=g) - 0.5)
    elif kind == "rough":         # large random steps: most candidates are invalid (slope / roughness), many full scans
        inp["measured_heights"] += 0.25 * (torch.rand(N, 693, generator=g) - 0.5) * (torch.rand(N, 1, generator=g) < 0.5)
    elif kind == "exceptions":    # heights more than 1 m off the base: the exception mask decides
        far = torch.rand(N, 693, generator=g) < 0.4
        inp["measured_heights"] = torch.where(far, inp["measured_heights"] + 3.0, inp["measured_heights"])
    elif kind == "flat":          # perfectly flat: slope 0 everywhere, exact ties resolved by the lowest index
        inp["measured_heights"] = (inp["root_states"][:, 2:3] - 0.3).expand(N, 693).contiguous()
    return {k: v.contiguous() for k, v in inp.items()}


@pytest.mark.parametrize("kind", ["border", "rough", "exceptions", "flat"])
def test_scorer_fast_kernel_stress_cases(kind):
    """Patches on / outside the grid border, envs without a valid candidate (grid-tiling fallback), exception masks and
    exact ties: fast kernel == oracle == generic kernel (debug call), bit for bit."""
    inp = _stress_inputs(kind)
    o, h = _scorer_case(inp, debug=False)
    _assert_scorer_equal(o, h, debug=False)
    _, hd = _scorer_case(inp, debug=True)
    for k in ("optimal_foothold_indice", "foothold_obs", "optimal_footholds_world", "pred_footholds", "pred_footholds_to_robot"):
        assert torch.equal(h[k], hd[k]), k
    if kind == "rough":           # the fallback really ran: some winners are not "valid" totals
        tot = torch.gather(hd["foothold_score"], 1, hd["optimal_foothold_indice"].expand(-1, 1, 4)).squeeze(1)
        assert int((tot >= 1.0).sum()) > 0


def test_scorer_fast_equals_generic_at_bench_size(monkeypatch):
    from dtc_amd import foothold
    big = {k: v.to(DEV) for k, v in S.scorer_inputs(98304, seed=7).items()}
    fast = foothold.plan(big["measured_heights"], big["root_states"], big["thigh_pos"], big["commands"])
    monkeypatch.setenv("DTC_PLANNER_GENERIC", "1")
    gen = foothold.plan(big["measured_heights"], big["root_states"], big["thigh_pos"], big["commands"])
    torch.cuda.synchronize()
    for k in ("optimal_foothold_indice", "foothold_obs", "optimal_footholds_world", "pred_footholds", "pred_footholds_to_robot"):
        assert torch.equal(fast[k], gen[k]), k


def test_scorer_edge_cases_bit_exact():
    from test_oracle_golden import scorer_edge_inputs
    o, h = _scorer_case(scorer_edge_inputs())
    _assert_scorer_equal(o, h)
    assert (_np(h["optimal_foothold_indice"])[0:8] == 0).all()
    o, h = _scorer_case(scorer_edge_inputs(), debug=False)        # the fast kernel on the same cases
    _assert_scorer_equal(o, h, debug=False)


def test_scorer_matches_reference_golden(golden):
    """HIP path against the fixture captured from the reference itself (knife-edge policy)."""
    from dtc_amd import foothold
    g = golden("scorer")
    inp = {k: v.to(DEV) for k, v in S.scorer_inputs(8192, seed=7).items()}
    h = foothold.plan(inp["measured_heights"], inp["root_states"], inp["thigh_pos"], inp["commands"])
    idx = _np(h["optimal_foothold_indice"]).squeeze(1)
    ref = g["main_idx"].astype(np.int64)
    for e, l in np.argwhere(idx != ref):
        assert g["main_gap"][e, l] <= 1e-5
    np.testing.assert_allclose(_np(h["pred_footholds"])[::4], g["main_pred"], rtol=0, atol=4e-6)


@pytest.mark.parametrize("tag", ["seed2", "bench", "slopes"])
def test_scorer_matches_reference_golden_more_draws(golden, tag):
    """HIP planner against further reference-captured draws: another seed, the bench distribution (every 24th of the
    98304 maps bench.py plans over) and smooth tilted terrain."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from cases import scorer_extra_inputs
    from dtc_amd import foothold
    g = golden("scorer")
    inp = {k: v.to(DEV) for k, v in scorer_extra_inputs(tag).items()}
    h = foothold.plan(inp["measured_heights"], inp["root_states"], inp["thigh_pos"], inp["commands"])
    idx = _np(h["optimal_foothold_indice"]).squeeze(1)
    ref = g[tag + "_idx"].astype(np.int64)
    mism = np.argwhere(idx != ref)
    for e, l in mism:
        assert g[tag + "_gap"][e, l] <= 1e-5
    assert len(mism) <= 4
    ok = np.ones(len(ref), bool)
    ok[mism[:, 0]] = False
    np.testing.assert_array_equal(_np(h["foothold_obs"])[::8][ok[::8]], g[tag + "_foothold_obs"][ok[::8]])
    np.testing.assert_allclose(_np(h["pred_footholds"])[::8], g[tag + "_pred"], rtol=0, atol=4e-6)


def test_scorer_unaligned_pointer_and_large():
    """A view that is not 16-byte aligned takes the scalar-load path; 98304 maps = 24 recorded steps."""
    from dtc_amd import foothold
    from oracle import foothold as OF
    inp = S.scorer_inputs(1024, seed=3)
    buf = torch.empty(1024 * 693 + 1, device=DEV)
    mh = buf[1:].view(1024, 693)
    mh.copy_(inp["measured_heights"])
    assert mh.data_ptr() % 16 != 0
    h = foothold.plan(mh, inp["root_states"].to(DEV), inp["thigh_pos"].to(DEV), inp["commands"].to(DEV))
    o = OF.plan(_np(inp["measured_heights"]), _np(inp["root_states"]), _np(inp["thigh_pos"]), _np(inp["commands"]),
                S.MEASURED_POINTS_X, S.MEASURED_POINTS_Y)
    np.testing.assert_array_equal(_np(h["optimal_foothold_indice"]).squeeze(1), o["idx"])
    # full bench size: property check (index decodes to the returned world position; idempotent)
    big = {k: v.to(DEV) for k, v in S.scorer_inputs(98304, seed=5).items()}
    h1 = foothold.plan(big["measured_heights"], big["root_states"], big["thigh_pos"], big["commands"])
    h2 = foothold.plan(big["measured_heights"], big["root_states"], big["thigh_pos"], big["commands"])
    assert torch.equal(h1["optimal_foothold_indice"], h2["optimal_foothold_indice"])
    idx = h1["optimal_foothold_indice"].squeeze(1)
    assert int(idx.min()) >= 0 and int(idx.max()) < 693
    z = torch.gather(big["measured_heights"], 1, idx)
    assert torch.equal(z, h1["optimal_footholds_world"][:, :, 2])


def test_get_heights_bit_exact():
    from dtc_amd import foothold
    from oracle import heights as OH
    tab = _terrain_table()
    inp = S.scorer_inputs(2048, seed=9)
    root = inp["root_states"]
    root[:8, 0] = torch.tensor([-30., -19.99, 0., 67.9, 68.0, 100., 20., 20.])
    root[:8, 1] = torch.tensor([-30., 0., -19.99, 35.9, 36.0, 100., -25., 40.])
    ref = OH.get_heights(tab.numpy(), root.numpy(), S.MEASURED_POINTS_X, S.MEASURED_POINTS_Y)
    got = foothold.get_heights(tab.to(DEV), root.to(DEV))
    np.testing.assert_array_equal(_np(got), ref)


def _terrain_table(seed=31):
    gen = torch.Generator().manual_seed(seed)
    coarse = torch.randint(-60, 120, (1760 // 16, 1120 // 16), generator=gen)
    tab = coarse.repeat_interleave(16, 0).repeat_interleave(16, 1)
    return (tab + torch.randint(-2, 3, (1760, 1120), generator=gen)).to(torch.int16)


@pytest.mark.parametrize("N,hscale", [(1, 0.05), (5, 0.05), (2050, 0.05), (1031, 0.1), (515, 0.0625)])
def test_plan_from_table_equals_get_heights_then_plan(N, hscale):
    """Row f1 fused into the planner: one launch samples the int16 terrain table, writes measured_heights and plans.
    Bit-identical to the two separate launches and to the oracle (heights + planner); 0.05 / 0.1 take the two-fma
    division, any other scale the IEEE one."""
    from dtc_amd import foothold
    from oracle import foothold as OF, heights as OH
    tab = _terrain_table()
    inp = S.scorer_inputs(N, seed=40 + N)
    root = inp["root_states"]
    k = min(8, N)
    root[:k, 0] = torch.tensor([-30., -19.99, 0., 67.9, 68.0, 100., 20., 20.])[:k]        # on / past the table border
    root[:k, 1] = torch.tensor([-30., 0., -19.99, 35.9, 36.0, 100., -25., 40.])[:k]
    root[:, 2] = 0.3 + 0.005 * 30                                                        # base ~0.3 m above the mid level
    d = {k2: v.to(DEV) for k2, v in inp.items()}
    fused = foothold.plan_from_table(tab.to(DEV), d["root_states"], d["thigh_pos"], d["commands"], horizontal_scale=hscale)
    mh = foothold.get_heights(tab.to(DEV), d["root_states"], horizontal_scale=hscale)
    two = foothold.plan(mh, d["root_states"], d["thigh_pos"], d["commands"])
    torch.cuda.synchronize()
    assert torch.equal(fused["measured_heights"], mh)
    for key in ("optimal_foothold_indice", "foothold_obs", "optimal_footholds_world", "pred_footholds", "pred_footholds_to_robot"):
        assert torch.equal(fused[key], two[key]), key
    ref_h = OH.get_heights(tab.numpy(), root.numpy(), S.MEASURED_POINTS_X, S.MEASURED_POINTS_Y, horizontal_scale=hscale)
    np.testing.assert_array_equal(_np(fused["measured_heights"]), ref_h)
    o = OF.plan(ref_h, root.numpy(), _np(inp["thigh_pos"]), _np(inp["commands"]), S.MEASURED_POINTS_X, S.MEASURED_POINTS_Y)
    np.testing.assert_array_equal(_np(fused["optimal_foothold_indice"]).squeeze(1), o["idx"])
    np.testing.assert_array_equal(_np(fused["optimal_footholds_world"]), o["optimal_footholds_world"])


def test_plan_from_table_other_grid_runs_the_two_launches():
    from dtc_amd import foothold
    grid = foothold.GridConfig(tuple(np.linspace(-0.4, 0.4, 17).astype(np.float32).tolist()),
                               tuple(np.linspace(-0.25, 0.25, 11).astype(np.float32).tolist()))
    tab = _terrain_table().to(DEV)
    inp = {k: v.to(DEV) for k, v in S.scorer_inputs(300, seed=3).items()}
    fused = foothold.plan_from_table(tab, inp["root_states"], inp["thigh_pos"], inp["commands"], grid=grid)
    mh = foothold.get_heights(tab, inp["root_states"], grid=grid)
    two = foothold.plan(mh, inp["root_states"], inp["thigh_pos"], inp["commands"], grid=grid)
    assert torch.equal(fused["measured_heights"], mh)
    assert torch.equal(fused["optimal_foothold_indice"], two["optimal_foothold_indice"])


def test_patch_env_hooks_write_the_reference_attributes():
    """`patch_env` on a stand-in env object: `plan_footholds()` (recorded heights) and `measure_and_plan_footholds()`
    (height field) set the attributes legged_robot_dtc.py:98-201 sets, with identical values."""
    from types import SimpleNamespace as NS
    from dtc_amd import foothold
    N = 260
    inp = {k: v.to(DEV) for k, v in S.scorer_inputs(N, seed=77).items()}
    rb = torch.zeros(N, 17, 13, device=DEV)
    thigh_indices = torch.tensor([2, 6, 10, 14], device=DEV)
    rb[:, thigh_indices, 0:3] = inp["thigh_pos"]
    tab = _terrain_table().to(DEV)
    env = NS(num_envs=N, num_bodies=17, rigid_body_state=rb.view(N, 17 * 13), thigh_indices=thigh_indices,
             root_states=inp["root_states"], commands=inp["commands"], height_samples=tab,
             terrain=NS(cfg=NS(border_size=20.0, horizontal_scale=0.05, vertical_scale=0.005)),
             cfg=NS(terrain=NS(measured_points_x=S.MEASURED_POINTS_X, measured_points_y=S.MEASURED_POINTS_Y),
                    sim=NS(dt=0.005), control=NS(decimation=4)))
    foothold.patch_env(env)
    env.measure_and_plan_footholds()
    fused_idx, fused_obs, mh = env.optimal_foothold_indice.clone(), env.foothold_obs.clone(), env.measured_heights
    assert mh.shape == (N, 693) and torch.equal(mh, foothold.get_heights(tab, inp["root_states"]))
    env.plan_footholds()
    assert torch.equal(env.optimal_foothold_indice, fused_idx) and torch.equal(env.foothold_obs, fused_obs)
    assert torch.equal(env.hip_positions, inp["thigh_pos"])
    for k in ("pred_footholds", "pred_footholds_to_robot", "optimal_footholds_world"):
        assert getattr(env, k).shape == (N, 4, 3)


# ------------------------------------------------------------------------------ GAE / gather
@pytest.mark.parametrize("N", [1, 64, 1000, 4096])
def test_gae_vs_oracle(N):
    from dtc_amd import ops
    from oracle import gae as OG
    d = S.rollout(N, 24, seed=4)
    d["last_values"] = torch.linspace(-1, 1, N).unsqueeze(1)
    d["dones"][0, : max(1, N // 8)] = 1
    d["dones"][23, : max(1, N // 4)] = 1
    sq = lambda k: _np(d[k].squeeze(-1))
    ret, adv = OG.compute_returns(sq("rewards"), sq("values"), sq("dones"), sq("last_values"))
    g = {k: d[k].to(DEV).contiguous() for k in ("rewards", "values", "dones", "last_values")}
    returns = torch.empty(24, N, 1, device=DEV)
    advantages = torch.empty(24, N, 1, device=DEV)
    stats = torch.zeros(4, dtype=torch.float64, device=DEV)
    ops.gae(g["rewards"], g["values"], g["dones"], g["last_values"], 0.99, 0.95, returns, advantages, stats)
    np.testing.assert_array_equal(_np(returns).squeeze(-1), ret)            # scan: bit exact
    raw = _np(advantages).squeeze(-1).copy()
    np.testing.assert_array_equal(raw, (ret - sq("values")).astype(np.float32))
    if N > 1:
        ops.adv_sqdev(advantages, stats, 24 * N)
        ops.adv_normalize(advantages, stats, 24 * N)
        np.testing.assert_allclose(_np(advantages).squeeze(-1), adv, rtol=2e-6, atol=2e-6)
        st = _np(stats)
        assert abs(st[0] - raw.astype(np.float64).sum()) <= 1e-9 * np.abs(raw).sum() + 1e-12


@pytest.mark.parametrize("shape,dtype", [((1000, 1389), torch.float32), ((1000, 53), torch.float32),
                                         ((1000, 12), torch.float32), ((1000, 1), torch.uint8),
                                         ((1000, 3), torch.float32), ((777, 265), torch.float32)])
def test_gather_rows(shape, dtype):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(1)
    src = (torch.randn(shape, generator=g) * 50).to(dtype).to(DEV)
    idx = torch.randperm(shape[0], generator=g)[: shape[0] // 2].to(DEV)
    out = ops.gather_rows(src, idx)
    assert torch.equal(out, src[idx])
    empty = ops.gather_rows(src, idx[:0])
    assert empty.shape[0] == 0


# ------------------------------------------------------------------------------ dense layers
def _ref_act(z, act):
    if act == "relu":
        return torch.relu(z)
    if act == "elu":
        return torch.nn.functional.elu(z)
    return z


@pytest.mark.parametrize("M,N,K,act", [(384, 512, 693, "relu"), (384, 512, 512, "elu"), (300, 64, 128, "relu"),
                                       (129, 35, 64, None), (384, 53, 128, None), (257, 12, 128, None),
                                       (384, 1, 128, None), (128, 693, 512, None), (1000, 256, 512, "elu"),
                                       (64, 128, 265, "relu")])
def test_linear_fwd_plain(M, N, K, act):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.randn(M, K, generator=g)
    W = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g)
    ref = _ref_act(X.double() @ W.double().t() + b.double(), act)
    Y = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_fwd(X.to(DEV), W.to(DEV), b.to(DEV), Y, act)
    np.testing.assert_allclose(_np(Y), ref.numpy(), rtol=2e-5, atol=2e-5)


def test_linear_fwd_detects_transposed_output():
    """A = I with an asymmetric W: catches a row/col swap in the MFMA C/D mapping."""
    from dtc_amd import ops
    K = 128
    X = torch.eye(K)
    W = torch.arange(K * K, dtype=torch.float32).view(K, K) / 100.0        # W[n,k] asymmetric
    Y = torch.zeros(K, K, device=DEV)
    ops.linear_fwd(X.to(DEV), W.to(DEV), None, Y, None)
    if ops.SPLIT and ops.H2:        # two fp16 terms carry 22 of the 24 significant bits of W (three bf16 terms: all of them)
        np.testing.assert_allclose(_np(Y), W.t().numpy(), rtol=2.0 ** -21, atol=0)
    else:
        np.testing.assert_array_equal(_np(Y), W.t().numpy())


def test_linear_fwd_segments_and_gather():
    """actor input = cat[obs[idx], z, mu[:, :3], l_t]  (actor_critic_decoder.py:431)."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(5)
    R, B = 2000, 384
    obs_all = torch.randn(R, 53, generator=g)
    idx = torch.randperm(R, generator=g)[:B]
    z = torch.randn(B, 16, generator=g)
    mulv = torch.randn(B, 35, generator=g)
    l_t = torch.randn(B, 512, generator=g)
    W = torch.randn(512, 584, generator=g) / 584 ** 0.5
    b = torch.randn(512, generator=g)
    cat = torch.cat([obs_all[idx], z, mulv[:, :3], l_t], dim=1)
    ref = torch.nn.functional.elu(cat.double() @ W.double().t() + b.double())
    d = lambda t: t.to(DEV)
    obs_d, z_d, mulv_d, lt_d, idx_d = d(obs_all), d(z), d(mulv), d(l_t), d(idx)
    X = _ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(z_d, 0, 16), _ffi.seg(mulv_d, 0, 3),
                     _ffi.seg(lt_d, 0, 512)], idx_d)
    Y = torch.empty(B, 512, device=DEV)
    ops.linear_fwd(X, d(W), d(b), Y, "elu")
    np.testing.assert_allclose(_np(Y), ref.numpy(), rtol=2e-5, atol=2e-5)
    # critic input = cat[obs[idx], base_vel[idx], priv[idx, 693:1389]]  (actor_critic_decoder.py:550)
    priv = torch.randn(R, 1389, generator=g)
    bv = torch.randn(R, 3, generator=g)
    Wc = torch.randn(512, 752, generator=g) / 752 ** 0.5
    catc = torch.cat([obs_all[idx], bv[idx], priv[idx, 693:]], dim=1)
    refc = torch.nn.functional.elu(catc.double() @ Wc.double().t() + b.double())
    priv_d, bv_d = d(priv), d(bv)
    Xc = _ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(bv_d, 0, 3, gather=True),
                      _ffi.seg(priv_d, 693, 696, gather=True)], idx_d)
    ops.linear_fwd(Xc, d(Wc), d(b), Y, "elu")
    np.testing.assert_allclose(_np(Y), refc.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("M,N,K,act", [(384, 512, 512, "relu"), (300, 693, 512, "relu"), (257, 12, 128, "elu"),
                                       (384, 1, 128, "elu"), (129, 64, 128, "relu"), (384, 128, 64, None),
                                       (384, 35, 64, None)])
def test_linear_dgrad_plain(M, N, K, act):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M * 3 + N + K)
    dZ = torch.randn(M, N, generator=g)
    W = torch.randn(N, K, generator=g) / N ** 0.5
    Xs = _ref_act(torch.randn(M, K, generator=g), act)
    ref = dZ.double() @ W.double()
    if act == "relu":
        ref = ref * (Xs > 0)
    elif act == "elu":
        ref = torch.where(Xs > 0, ref, ref * (Xs.double() + 1.0))
    dX = torch.full((M, K), float("nan"), device=DEV)
    ops.linear_dgrad(dZ.to(DEV), W.to(DEV), dX, Xs.to(DEV) if act else None, act)
    np.testing.assert_allclose(_np(dX), ref.numpy(), rtol=2e-5, atol=2e-5)


def test_linear_dgrad_segments_accumulate():
    """cenet_decoder layer 0: d[z | mu[:, :3] | l_t] with accumulation into existing gradients."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(8)
    B = 384
    dZ = torch.randn(B, 64, generator=g)
    W = torch.randn(64, 531, generator=g) / 8
    dz0, dmulv0, dlt0 = torch.randn(B, 16, generator=g), torch.randn(B, 35, generator=g), torch.randn(B, 512, generator=g)
    full = dZ.double() @ W.double()
    dz, dmulv, dlt = dz0.to(DEV), dmulv0.to(DEV), dlt0.to(DEV)
    dst = _ffi.segmat([_ffi.seg(dz, 0, 16), _ffi.seg(dmulv, 0, 3, accumulate=True),
                       _ffi.seg(dlt, 0, 512, accumulate=True)])
    ops.linear_dgrad(dZ.to(DEV), W.to(DEV), dst)
    np.testing.assert_allclose(_np(dz), full[:, :16].numpy(), rtol=2e-5, atol=2e-5)
    exp_mulv = dmulv0.double().clone()
    exp_mulv[:, :3] += full[:, 16:19]
    np.testing.assert_allclose(_np(dmulv), exp_mulv.numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(_np(dlt), (dlt0.double() + full[:, 19:]).numpy(), rtol=2e-5, atol=2e-5)
    # null segment (obs needs no gradient): actor layer 0
    Wa = torch.randn(64, 584, generator=g) / 8
    fulla = dZ.double() @ Wa.double()
    dz2, dmulv2, dlt2 = torch.zeros(B, 16, device=DEV), torch.zeros(B, 35, device=DEV), torch.zeros(B, 512, device=DEV)
    dsta = _ffi.segmat([_ffi.seg(None, 0, 53), _ffi.seg(dz2, 0, 16), _ffi.seg(dmulv2, 0, 3), _ffi.seg(dlt2, 0, 512)])
    ops.linear_dgrad(dZ.to(DEV), Wa.to(DEV), dsta)
    np.testing.assert_allclose(_np(dz2), fulla[:, 53:69].numpy(), rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(_np(dmulv2)[:, :3], fulla[:, 69:72].numpy(), rtol=2e-5, atol=2e-5)
    assert float(dmulv2[:, 3:].abs().max()) == 0.0
    np.testing.assert_allclose(_np(dlt2), fulla[:, 72:].numpy(), rtol=2e-5, atol=2e-5)

@pytest.mark.parametrize("M,N,K,Kd", [(24576, 512, 693, 512), (24576, 512, 512, 693), (384, 128, 265, 64), (24576, 64, 531, 128),
                                      (256, 64, 128, 35), (4096, 128, 64, 53), (1024, 512, 300, 512)])
def test_relu_sign_record_fwd_and_dgrad_are_bit_identical(M, N, K, Kd):
    """dtc_linear_fwd_mask writes the same Y as dtc_linear_fwd(relu) plus one bit per element; dtc_linear_dgrad_mask on
    that record gives bit for bit the data gradient dtc_linear_dgrad computes from the saved activation -- for every tile
    shape the launches pick (128-, 64- and 32-wide tiles, the two-stages-ahead variants of small grids).  Layer under
    test: Y = relu(X W^T + b) [M, N]; the NEXT layer's data gradient dY = (dZ [M, Kd] Wn [Kd, N]) * (Y > 0)."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + 7 * N + K)
    X = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    assert ops.relu_mask_ok(M, N)
    Y0, Y1 = torch.empty(M, N, device=DEV), torch.empty(M, N, device=DEV)
    mask = ops.relu_mask(M, N, DEV)
    mask.fill_(0x5555)
    ops.linear_fwd(X, W, b, Y0, "relu")
    ops.linear_fwd(X, W, b, Y1, "relu", mask=mask)
    assert torch.equal(Y0, Y1)
    # the record itself: bit r of word [(row // 32) * 2 + half][col] <-> row 32 * blk + 4 * half + (r & 3) + 8 * (r >> 2)
    pos = (Y0 > 0).cpu().view(M // 32, 4, 2, 4, N)                       # [blk, r >> 2, half, r & 3, col]
    bits = pos.permute(0, 2, 1, 3, 4).reshape(M // 32, 2, 16, N).to(torch.int32)
    want = (bits << torch.arange(16, dtype=torch.int32).view(1, 1, 16, 1)).sum(dim=2).reshape(-1, N)
    got = mask.cpu().view(-1, N).to(torch.int32) & 0xFFFF
    assert torch.equal(got, want)
    dZ = torch.randn(M, Kd, generator=g).to(DEV)
    Wn = (torch.randn(Kd, N, generator=g) / Kd ** 0.5).to(DEV)
    d0 = torch.full((M, N), float("nan"), device=DEV)
    d1 = torch.full((M, N), float("nan"), device=DEV)
    ops.linear_dgrad(dZ, Wn, d0, Y0, "relu")
    ops.linear_dgrad(dZ, Wn, d1, Y0, "relu", mask=mask)
    assert torch.equal(d0, d1)
    ref = (dZ.double() @ Wn.double()) * (Y0 > 0)
    np.testing.assert_allclose(_np(d1), ref.cpu().numpy(), rtol=2e-5, atol=2e-5)
    # accumulating destination
    from dtc_amd import _ffi
    base = torch.randn(M, N, generator=g).to(DEV)
    a0, a1 = base.clone(), base.clone()
    ops.linear_dgrad(dZ, Wn, _ffi.segmat([_ffi.seg(a0, 0, N, accumulate=True)]), Y0, "relu")
    ops.linear_dgrad(dZ, Wn, _ffi.segmat([_ffi.seg(a1, 0, N, accumulate=True)]), Y0, "relu", mask=mask)
    assert torch.equal(a0, a1)


def test_relu_sign_record_rejects_ineligible_shapes():
    from dtc_amd import _ffi, ops
    X, W, Y = torch.randn(200, 64, device=DEV), torch.randn(96, 64, device=DEV), torch.empty(200, 96, device=DEV)
    assert not ops.relu_mask_ok(200, 96)
    with pytest.raises(_ffi.DtcError, match="sign record"):
        ops.linear_fwd(X, W, None, Y, "relu", mask=torch.empty(4096, dtype=torch.int16, device=DEV))


@pytest.mark.parametrize("M,N,K", [(24576, 128, 256), (24576, 64, 531), (24576, 35, 64), (24576, 1, 128), (4096, 512, 752),
                                   (4096, 256, 512), (1000, 53, 128), (24576, 12, 128)])
def test_deep_prefetch_variants_are_bit_identical(M, N, K, monkeypatch):
    """Small launches (at most 300 workgroups by default) take the two-stages-ahead kernels (csrc/gemm.hip, DEEP): same operand images, same
    k order, same accumulation -- forward and data gradient must equal the single-stage kernels bit for bit."""
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.randn(M, K, generator=g).to(DEV)
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(DEV)
    b = torch.randn(N, generator=g).to(DEV)
    dZ = torch.randn(M, N, generator=g).to(DEV)
    outs = []
    for thr in ("0", "100000"):
        monkeypatch.setenv("DTC_GEMM_DEEP_BLOCKS", thr)
        Y = torch.empty(M, N, device=DEV)
        ops.linear_fwd(X, W, b, Y, act="elu")
        dX = torch.empty(M, K, device=DEV)
        ops.linear_dgrad(dZ, W, dX, Xsaved=X, act="elu")
        torch.cuda.synchronize()
        outs.append((Y, dX))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    ref = torch.nn.functional.elu(X.double() @ W.double().t() + b.double())
    assert float((outs[1][0].double() - ref).abs().max()) < 2e-4 * max(1.0, float(ref.abs().max()))



@pytest.mark.parametrize("act", ["relu", "crelu", "elu", "selu", "lrelu", "tanh", "sigmoid", None])
@pytest.mark.parametrize("M,N,K", [(384, 128, 265), (1000, 70, 100)])
def test_linear_activations_fwd_and_dgrad_vs_torch(act, M, N, K):
    """Every entry of the reference's get_activation table (actor_critic_decoder.py:565-582): forward epilogue and the
    derivative the data-gradient epilogue takes from the saved post-activation output, against torch autograd."""
    from dtc_amd import ops
    from dtc_amd.modules.actor_critic_decoder import get_activation
    g = torch.Generator().manual_seed(M + N + K)
    X = torch.randn(M, K, generator=g, dtype=torch.float64, requires_grad=True)
    W = torch.randn(N, K, generator=g, dtype=torch.float64) / K ** 0.5
    b = torch.randn(N, generator=g, dtype=torch.float64) * 0.3
    W2 = torch.randn(32, N, generator=g, dtype=torch.float64) / N ** 0.5
    dZ2 = torch.randn(M, 32, generator=g, dtype=torch.float64)
    fn = get_activation(act) if act is not None else (lambda t: t)
    Y = fn(X @ W.t() + b)
    (Y @ W2.t() * dZ2).sum().backward()              # d/dY = dZ2 W2 ; then through the activation and the layer
    f = lambda t: t.detach().float().to(DEV)
    Yh = torch.empty(M, N, device=DEV)
    ops.linear_fwd(f(X), f(W), f(b), Yh, act)
    np.testing.assert_allclose(_np(Yh), Y.detach().numpy(), rtol=2e-5, atol=2e-5)
    # backward of the NEXT layer with this layer's activation derivative fused: dY_pre = (dZ2 W2) * act'(Y)
    dpre = torch.empty(M, N, device=DEV)
    ops.linear_dgrad(f(dZ2), f(W2), dpre, Yh, act)
    dX = torch.empty(M, K, device=DEV)
    ops.linear_dgrad(dpre, f(W), dX, None, None)
    np.testing.assert_allclose(_np(dX), X.grad.numpy(), rtol=3e-4, atol=3e-5)


@pytest.mark.parametrize("M,N,K", [(1536, 512, 693), (1536, 512, 512), (1000, 64, 128), (384, 35, 64),
                                   (1536, 693, 512), (777, 12, 128), (384, 1, 128), (24576, 128, 256)])
def test_linear_wgrad_plain(M, N, K):
    from dtc_amd import ops
    g = torch.Generator().manual_seed(M + 7 * N + K)
    dZ = torch.randn(M, N, generator=g) / M ** 0.5
    X = torch.randn(M, K, generator=g)
    ref_w = dZ.double().t() @ X.double()
    ref_b = dZ.double().sum(0)
    dW = torch.full((N, K), float("nan"), device=DEV)
    db = torch.full((N,), float("nan"), device=DEV)
    ws = torch.empty(ops.wgrad_workspace_bytes(M, N, K) // 4, device=DEV)
    ops.linear_wgrad(dZ.to(DEV), X.to(DEV), dW, db, ws)
    np.testing.assert_allclose(_np(dW), ref_w.numpy(), rtol=3e-5, atol=3e-5)
    np.testing.assert_allclose(_np(db), ref_b.numpy(), rtol=3e-5, atol=3e-5)


def test_linear_wgrad_segments_and_gather():
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(15)
    R, B = 3000, 1536
    obs_all = torch.randn(R, 53, generator=g)
    priv = torch.randn(R, 1389, generator=g)
    bv = torch.randn(R, 3, generator=g)
    idx = torch.randperm(R, generator=g)[:B]
    dZ = torch.randn(B, 512, generator=g) / B ** 0.5
    cat = torch.cat([obs_all[idx], bv[idx], priv[idx, 693:]], dim=1)
    ref_w = dZ.double().t() @ cat.double()
    d = lambda t: t.to(DEV)
    obs_d, bv_d, priv_d, idx_d = d(obs_all), d(bv), d(priv), d(idx)
    X = _ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(bv_d, 0, 3, gather=True),
                     _ffi.seg(priv_d, 693, 696, gather=True)], idx_d)
    dW = torch.empty(512, 752, device=DEV)
    db = torch.empty(512, device=DEV)
    ws = torch.empty(ops.wgrad_workspace_bytes(B, 512, 752) // 4, device=DEV)
    ops.linear_wgrad(d(dZ), X, dW, db, ws)
    np.testing.assert_allclose(_np(dW), ref_w.numpy(), rtol=3e-5, atol=3e-5)
    np.testing.assert_allclose(_np(db), dZ.double().sum(0).numpy(), rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("M", [1536, 5000])
def test_wgrad_group_matches_fp64(M):
    """dtc_wgrad_group: the layers of one bucket (wide, narrow, 1-row, segmented + gathered) in one launch pair, against
    the fp64 products; a second call on the same workspace must reproduce the first bit for bit."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(100 + M)
    R = M + 700
    d = lambda t: t.to(DEV)
    obs_all, priv, bv = torch.randn(R, 53, generator=g), torch.randn(R, 1389, generator=g), torch.randn(R, 3, generator=g)
    idx = torch.randperm(R, generator=g)[:M]
    obs_d, priv_d, bv_d, idx_d = d(obs_all), d(priv), d(bv), d(idx)
    shapes = [(512, 752), (512, 512), (256, 512), (128, 256), (12, 128), (1, 128), (35, 64), (693, 512)]
    jobs, refs = [], []
    for li, (N, K) in enumerate(shapes):
        dZ = torch.randn(M, N, generator=g) / M ** 0.5
        if li == 0:
            Xh = torch.cat([obs_all[idx], bv[idx], priv[idx, 693:]], dim=1)
            X = _ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(bv_d, 0, 3, gather=True),
                             _ffi.seg(priv_d, 693, 696, gather=True)], idx_d)
        else:
            Xh = torch.randn(M, K, generator=g)
            X = d(Xh)
        dW = torch.full((N, K), float("nan"), device=DEV)
        db = torch.full((N,), float("nan"), device=DEV) if li != 3 else None
        jobs.append((d(dZ), X, dW, db))
        refs.append((dZ.double().t() @ Xh.double(), dZ.double().sum(0)))
    ws = ops.workspace(ops.wgrad_group_workspace_bytes(jobs, M), DEV)
    ops.wgrad_group(jobs, M, ws)
    first = [(j[2].clone(), None if j[3] is None else j[3].clone()) for j in jobs]
    for (dZ, X, dW, db), (rw, rb) in zip(jobs, refs):
        np.testing.assert_allclose(_np(dW), rw.numpy(), rtol=3e-5, atol=3e-5)
        if db is not None:
            np.testing.assert_allclose(_np(db), rb.numpy(), rtol=3e-5, atol=3e-5)
    ops.wgrad_group(jobs, M, ws)
    for (dZ, X, dW, db), (w0, b0) in zip(jobs, first):
        assert torch.equal(dW, w0) and (db is None or torch.equal(db, b0))
    with pytest.raises(_ffi.DtcError):
        ops.wgrad_group(jobs * 2, M, ws)            # more than 12 jobs per launch


def test_foothold_rewards_vs_oracle():
    from dtc_amd import foothold
    from oracle import foothold as OF
    from test_oracle_golden import _reward_case
    foot, world, contact = _reward_case()
    tr, miss = OF.rewards(foot.numpy(), world.numpy(), contact.numpy())
    t, m = foothold.rewards(foot.to(DEV), world.to(DEV), contact.to(DEV))
    np.testing.assert_allclose(_np(t), tr, rtol=2e-6, atol=2e-6)
    np.testing.assert_array_equal(_np(m), miss)


def test_pack_cols_equals_cat_of_gathered_blocks():
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(12)
    R, B = 5000, 1537
    obs, bv, z, mulv = torch.randn(R, 53, generator=g), torch.randn(R, 3, generator=g), torch.randn(B, 16, generator=g), torch.randn(B, 35, generator=g)
    idx = torch.randint(0, R, (B,), generator=g)
    d = lambda t: t.to(DEV)
    obs_d, bv_d, z_d, mulv_d, idx_d = d(obs), d(bv), d(z), d(mulv), d(idx)
    out = torch.full((B, 72), float("nan"), device=DEV)
    ops.pack_cols(_ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(z_d, 0, 16), _ffi.seg(mulv_d, 0, 3)], idx_d), out)
    assert torch.equal(out.cpu(), torch.cat([obs[idx], z, mulv[:, :3]], dim=1))
    out2 = torch.full((B, 60), float("nan"), device=DEV)            # row stride wider than the packed block
    ops.pack_cols(_ffi.segmat([_ffi.seg(obs_d, 0, 53, gather=True), _ffi.seg(bv_d, 0, 3, gather=True)], idx_d), out2)
    assert torch.equal(out2[:, :56].cpu(), torch.cat([obs[idx], bv[idx]], dim=1)) and torch.isnan(out2[:, 56:]).all()


@pytest.mark.gpu
def test_lr_adapt_rule_consumed_slot_and_nan_kl():
    """dtc_lr_adapt (ppo.py:301-307 for data-parallel callers): the three branches; a KL that is NaN itself leaves the learning rate
    unchanged (both comparisons of the reference are false); the slot is consumed -- a second call without a fresh deposit poisons lr."""
    from dtc_amd import ops
    dev = "cuda:0"
    lr = torch.tensor([1e-3], dtype=torch.float64, device=dev)
    for kl, want in ((0.05, 1e-3 / 1.5), (0.001, 1e-3), (0.012, 1e-3), (float("nan"), 1e-3), (-1.0, 1e-3)):
        lr.fill_(1e-3)
        slot = torch.tensor([kl], dtype=torch.float32, device=dev)
        ops.lr_adapt(slot, lr, 0.01)
        exp = want if kl != 0.001 else 1e-3 * 1.5
        assert abs(float(lr.item()) - exp) <= 1e-15, (kl, float(lr.item()), exp)
        assert int(slot.view(torch.int32).item()) == 0x7fc0dead           # consumed
    ops.lr_adapt(slot, lr, 0.01)                                          # nothing deposited since
    assert float(lr.item()) != float(lr.item())
