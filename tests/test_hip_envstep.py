"""Env-step kernels either side of the planner / rollout store (rows f2, f3) through the C ABI vs the oracle:
compute_observations + check_termination (bit-exact), the fused transition store (incl. the time-out bootstrap and a
broadcast sigma), the history roll (ping-pong semantics of HistoryWrapper).  GPU only."""
import numpy as np
import pytest
import torch

from dtc_amd import foothold, ops, synthetic as S
from oracle import observations as OO

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("N", [1, 1024, 4099])
def test_compute_observations_bit_exact(N):
    s = S.env_state(N, seed=13)
    ref_obs, ref_priv, ref_h = OO.compute_observations({k: v.numpy() for k, v in s.items()})
    d = {k: v.to(DEV) for k, v in s.items()}
    out = foothold.compute_observations(d["base_ang_vel"], d["projected_gravity"], d["commands"], d["dof_pos"],
                                        d["default_dof_pos"], d["dof_vel"], d["actions"], d["foothold_obs"],
                                        d["root_states"], d["measured_heights"], d["forces"], d["height_noise_offset"],
                                        d["u_obs"], d["noise_scale_vec"], d["u_heights"])
    np.testing.assert_array_equal(out["obs_buf"].cpu().numpy(), ref_obs)
    np.testing.assert_array_equal(out["privileged_obs_buf"].cpu().numpy(), ref_priv)
    np.testing.assert_array_equal(out["heights"].cpu().numpy(), ref_h)
    # noise-free variant (add_noise = False, no height noise)
    out = foothold.compute_observations(d["base_ang_vel"], d["projected_gravity"], d["commands"], d["dof_pos"],
                                        d["default_dof_pos"], d["dof_vel"], d["actions"], d["foothold_obs"],
                                        d["root_states"], d["measured_heights"], d["forces"])
    sn = {k: v.numpy() for k, v in s.items()}
    obs0, _, h0 = OO.compute_observations(sn, add_noise=False)
    np.testing.assert_array_equal(out["obs_buf"].cpu().numpy(), obs0)
    np.testing.assert_array_equal(out["privileged_obs_buf"][:, :693].cpu().numpy(), h0)
    np.testing.assert_array_equal(out["privileged_obs_buf"][:, 696:].cpu().numpy(), h0)


@pytest.mark.parametrize("N", [3, 1024])
def test_check_termination_bit_exact(N):
    s = S.env_state(N, seed=13)
    ref_reset, ref_to, ref_mean = OO.check_termination({k: v.numpy() for k, v in s.items()}, 1000)
    d = {k: v.to(DEV) for k, v in s.items()}
    reset, tout, mean = foothold.check_termination(d["contact_forces"], d["termination_contact_indices"],
                                                   d["episode_length_buf"], 1000, d["projected_gravity"],
                                                   d["root_states"], d["measured_heights"])
    np.testing.assert_array_equal(mean.cpu().numpy(), ref_mean)
    np.testing.assert_array_equal(reset.cpu().numpy(), ref_reset)
    np.testing.assert_array_equal(tout.cpu().numpy(), ref_to)


def test_store_transition_matches_copy_loop():
    from dtc_amd.storage import RolloutStorage
    N, T = 777, 3
    g = torch.Generator().manual_seed(1)
    st = RolloutStorage(N, T, [53], [1389], [265], [12], DEV)
    ref = RolloutStorage(N, T, [53], [1389], [265], [12], "cpu")
    for t in range(T):
        tr, trc = RolloutStorage.Transition(), RolloutStorage.Transition()
        fields = dict(observations=53, next_observations=53, privileged_observations=1389, observation_histories=265,
                      actions=12, values=1, action_mean=12, base_vel=3)
        for k, w in fields.items():
            v = torch.randn(N, w, generator=g)
            setattr(trc, k, v)
            setattr(tr, k, v.to(DEV))
        sig = torch.rand(12, generator=g) + 0.5
        trc.action_sigma, tr.action_sigma = sig.expand(N, 12), sig.to(DEV).expand(N, 12)         # row stride 0
        lp, rw = torch.randn(N, generator=g), torch.randn(N, generator=g)
        dn = torch.rand(N, generator=g) < 0.1
        to = torch.rand(N, generator=g) < 0.2
        trc.actions_log_prob, tr.actions_log_prob = lp, lp.to(DEV)
        trc.rewards, tr.rewards = rw.clone(), rw.to(DEV)
        trc.dones, tr.dones = dn, dn.to(DEV)
        if t == 1:
            ref.add_transitions(trc)
            st.add_transitions(tr)
        else:
            ref.add_transitions(trc, time_outs=to, gamma=0.99)
            st.add_transitions(tr, time_outs=to.to(DEV), gamma=0.99)
    for k in ("observations", "next_observations", "privileged_observations", "observation_histories", "actions", "rewards",
              "dones", "values", "actions_log_prob", "mu", "sigma", "base_vel"):
        assert torch.equal(getattr(st, k).cpu(), getattr(ref, k)), k
    assert st.step == T
    with pytest.raises(AssertionError):
        st.add_transitions(tr)


def test_history_roll_and_wrapper_pingpong():
    from dtc_amd.env import HistoryWrapper, ReplayEnv
    N, L, D = 1000, 5, 53
    g = torch.Generator().manual_seed(2)
    hist = torch.randn(N, L * D, generator=g)
    obs = torch.randn(N, D, generator=g)
    want = torch.cat((hist[:, D:], obs), dim=-1)
    h, o = hist.to(DEV), obs.to(DEV)
    out = torch.empty_like(h)
    ops.history_roll(h, o, out, L)
    assert torch.equal(out.cpu(), want) and torch.equal(h.cpu(), hist)
    ops.history_roll(h, o, h, L)                                   # aliasing is allowed
    assert torch.equal(h.cpu(), want)
    reset = (torch.rand(N, generator=g) < 0.3).to(torch.uint8)
    h2 = hist.to(DEV)
    ops.history_roll(h2, o, out, L, reset=reset.to(DEV))
    want2 = want.clone()
    want2[reset.bool(), :-D] = 0
    assert torch.equal(out.cpu(), want2)
    # wrapper: the tensor handed out at step t must still hold step t's history after the env stepped again
    env = HistoryWrapper(ReplayEnv(64, DEV))
    od = env.get_observations()
    first = od["obs_history"]
    snap = first.clone()
    od2, *_ = env.step(torch.zeros(64, 12, device=DEV))
    assert torch.equal(first, snap) and od2["obs_history"].data_ptr() != first.data_ptr()
    assert torch.equal(od2["obs_history"][:, :-53], snap[:, 53:]) and torch.equal(od2["obs_history"][:, -53:], od2["obs"])


def test_zero_sized_and_invalid_calls():
    """N = 0 is a no-op for every env-step entry point; invalid descriptors are rejected with DtcError (nothing is
    launched, nothing wraps around)."""
    from dtc_amd import _ffi
    s = {k: v.to(DEV) for k, v in S.env_state(4, seed=13).items()}
    e = lambda *shape, dt=torch.float32: torch.empty(*shape, dtype=dt, device=DEV)
    out = foothold.compute_observations(e(0, 3), e(0, 3), e(0, 4), e(0, 12), s["default_dof_pos"], e(0, 12), e(0, 12), e(0, 8),
                                        e(0, 13), e(0, 693), e(0, 17, 3))
    assert out["obs_buf"].shape == (0, 53) and out["privileged_obs_buf"].shape == (0, 1389)
    h = foothold.plan(e(0, 693), e(0, 13), e(0, 4, 3), e(0, 4))
    assert h["optimal_foothold_indice"].shape[0] == 0
    with pytest.raises(_ffi.DtcError):
        ops.history_roll(e(8, 2000), e(8, 400), e(8, 2000), 5)                       # row longer than 1024 floats
    cfg = foothold.ObsConfig(term_row0=600, term_row1=500)
    with pytest.raises(_ffi.DtcError):
        foothold.check_termination(s["contact_forces"], s["termination_contact_indices"], s["episode_length_buf"], 1000,
                                   s["projected_gravity"], s["root_states"], s["measured_heights"], cfg)
    X, W, b = e(64, 512), e(693, 512), e(693)
    tgt, idx = e(100, 1389), torch.zeros(64, dtype=torch.int64, device=DEV)
    with pytest.raises(_ffi.DtcError):                                              # target columns outside the row
        _ffi.check(_ffi.lib().dtc_linear_fwd_mse(ops.as_segmat(X), W.data_ptr(), b.data_ptr(), tgt.data_ptr(), 1389, 100, 1000,
                                                 idx.data_ptr(), 1.0, e(64, 693).data_ptr(), 693,
                                                 torch.zeros(64, dtype=torch.float64, device=DEV).data_ptr(), 64, 693, 512,
                                                 torch.cuda.current_stream().cuda_stream), "dtc_linear_fwd_mse")


def _terrain_table(seed=31):
    gen = torch.Generator().manual_seed(seed)
    coarse = torch.randint(-60, 120, (1760 // 16, 1120 // 16), generator=gen)
    tab = coarse.repeat_interleave(16, 0).repeat_interleave(16, 1)
    return (tab + torch.randint(-2, 3, (1760, 1120), generator=gen)).to(torch.int16)


def _env_step_inputs(N, seed):
    s = S.env_state(N, seed=seed)
    g = torch.Generator().manual_seed(seed + 77)
    s["foot_positions"] = torch.cat([s["root_states"][:, None, :2] + 0.3 * torch.randn(N, 4, 2, generator=g),
                                     0.05 * torch.randn(N, 4, 1, generator=g)], dim=2)
    s["contact_filt"] = torch.rand(N, 4, generator=g) < 0.6
    return s


def _separate(d, mh_or_table, table, noise):
    """the four (five) stand-alone launches in the reference's order"""
    if table:
        p = foothold.plan_from_table(mh_or_table, d["root_states"], d["thigh_pos"], d["commands"])
        mh = p["measured_heights"]
    else:
        p = foothold.plan(mh_or_table, d["root_states"], d["thigh_pos"], d["commands"])
        mh = mh_or_table
    reset, tout, mean = foothold.check_termination(d["contact_forces"], d["termination_contact_indices"], d["episode_length_buf"],
                                                   1000, d["projected_gravity"], d["root_states"], mh)
    tr, miss = foothold.rewards(d["foot_positions"], p["optimal_footholds_world"], d["contact_filt"])
    nz = (d["height_noise_offset"], d["u_obs"], d["noise_scale_vec"], d["u_heights"]) if noise else (None,) * 4
    ob = foothold.compute_observations(d["base_ang_vel"], d["projected_gravity"], d["commands"], d["dof_pos"], d["default_dof_pos"],
                                       d["dof_vel"], d["actions"], p["foothold_obs"], d["root_states"], mh, d["forces"], *nz)
    out = dict(p)
    out.update(measured_heights=mh, reset_buf=reset, time_out_buf=tout, height_mean=mean, rew_tracking_optimal_footholds=tr,
               rew_foothold_miss=miss, **ob)
    return out


def _fused_kwargs(d, noise):
    kw = {k: d[k] for k in ("root_states", "thigh_pos", "commands", "contact_forces", "termination_contact_indices",
                            "episode_length_buf", "projected_gravity", "foot_positions", "contact_filt", "base_ang_vel", "dof_pos",
                            "default_dof_pos", "dof_vel", "actions", "forces")}
    if noise:
        kw.update({k: d[k] for k in ("height_noise_offset", "u_obs", "noise_scale_vec", "u_heights")})
    return kw


@pytest.mark.parametrize("N,table,noise", [(1, False, True), (4096, False, True), (1021, False, False), (4096, True, True), (1027, True, False), (18, True, True)])
def test_env_post_physics_one_launch_equals_the_separate_launches(N, table, noise):
    """configs[3] (one env step, 4096 envs): `foothold.EnvStep` = [heights from the terrain table +] foothold block + check_termination
    + foothold rewards + compute_observations in ONE launch.  Every output equals the separate launches bit for bit (each of which is
    pinned bit-exact to the oracle above / in test_hip_kernels.py), ragged sizes and both height sources included; the observation
    and termination outputs are also compared with the oracle directly."""
    s = _env_step_inputs(N, seed=13)
    if table:
        s["root_states"][:, 2] = 0.3 + 0.005 * 30
        s["root_states"][N // 32: N // 16 + 1, 2] -= 0.25
    d = {k: v.to(DEV) for k, v in s.items()}
    src = _terrain_table().to(DEV) if table else d["measured_heights"]
    want = _separate(d, src, table, noise)
    step = foothold.EnvStep(N, DEV)
    got = step(max_episode_length=1000, **({"height_samples": src} if table else {"measured_heights": src}), **_fused_kwargs(d, noise))
    torch.cuda.synchronize()
    for k, w in want.items():
        g = got[k]
        if w.dtype == torch.bool:
            g = g.bool()
        assert g.shape == w.shape and torch.equal(g, w), k
    assert 0 < int(got["reset_buf"].sum()) < N or N < 8
    # oracle, directly (the planner part is covered against the oracle in test_hip_kernels.py)
    sn = {k: v.cpu().numpy() for k, v in d.items()}
    sn["measured_heights"] = got["measured_heights"].cpu().numpy()
    sn["foothold_obs"] = got["foothold_obs"].cpu().numpy()
    ref_reset, ref_to, ref_mean = OO.check_termination(sn, 1000)
    np.testing.assert_array_equal(got["reset_buf"].bool().cpu().numpy(), ref_reset)
    np.testing.assert_array_equal(got["time_out_buf"].bool().cpu().numpy(), ref_to)
    np.testing.assert_array_equal(got["height_mean"].cpu().numpy(), ref_mean)
    ref_obs, ref_priv, ref_h = OO.compute_observations(sn, add_noise=noise)
    np.testing.assert_array_equal(got["obs_buf"].cpu().numpy(), ref_obs)
    np.testing.assert_array_equal(got["heights"].cpu().numpy(), ref_h)
    if noise:
        np.testing.assert_array_equal(got["privileged_obs_buf"].cpu().numpy(), ref_priv)
    # a second call on the same object reuses its buffers and gives the same bits
    again = step(max_episode_length=1000, **({"height_samples": src} if table else {"measured_heights": src}), **_fused_kwargs(d, noise))
    assert again["obs_buf"].data_ptr() == got["obs_buf"].data_ptr() and torch.equal(again["obs_buf"], want["obs_buf"])


def test_observation_rows_of_reset_envs_are_refreshed_in_place():
    """The reference resets envs between the rewards and the observations (legged_robot_dtc.py:209-211): after `EnvStep`, the
    caller resets and recomputes ONLY the reset envs' rows (`compute_observations(where=reset_buf, out=...)`); the result equals a
    full observation pass over the post-reset state, and the rows of the other envs are not touched."""
    N = 2048
    s = _env_step_inputs(N, seed=21)
    d = {k: v.to(DEV) for k, v in s.items()}
    step = foothold.EnvStep(N, DEV)
    kw = _fused_kwargs(d, True)
    got = step(max_episode_length=1000, measured_heights=d["measured_heights"], **kw)
    reset = got["reset_buf"].bool()
    assert 0 < int(reset.sum()) < N
    before = {k: got[k].clone() for k in ("obs_buf", "privileged_obs_buf", "heights")}
    post = dict(d)                                          # reset_idx: new joint / base state for the reset envs only
    g = torch.Generator(device=DEV).manual_seed(5)
    for k in ("dof_pos", "dof_vel", "actions", "base_ang_vel", "root_states"):
        fresh = torch.randn(d[k].shape, generator=g, device=DEV)
        post[k] = torch.where(reset.view(-1, *[1] * (d[k].dim() - 1)), fresh, d[k])
    args = lambda q: (q["base_ang_vel"], q["projected_gravity"], q["commands"], q["dof_pos"], q["default_dof_pos"], q["dof_vel"],   # noqa: E731
                      q["actions"], got["foothold_obs"], q["root_states"], d["measured_heights"], q["forces"], q["height_noise_offset"],
                      q["u_obs"], q["noise_scale_vec"], q["u_heights"])
    full = foothold.compute_observations(*args(post))
    foothold.compute_observations(*args(post), where=reset, out=got)
    for k in before:
        assert torch.equal(got[k], full[k]), k
        assert torch.equal(got[k][~reset], before[k][~reset]), k
    with pytest.raises(ValueError):
        foothold.compute_observations(*args(post), where=reset)


def test_env_post_physics_other_grid_and_bad_arguments():
    from dtc_amd import _ffi
    grid = foothold.GridConfig(tuple(np.linspace(-0.4, 0.4, 17).astype(np.float32).tolist()),
                               tuple(np.linspace(-0.25, 0.25, 11).astype(np.float32).tolist()))
    cfg = foothold.ObsConfig(num_points=17 * 11, term_row0=3 * 11, term_row1=14 * 11)
    N = 300
    s = _env_step_inputs(N, seed=3)
    d = {k: v.to(DEV) for k, v in s.items()}
    mh = d["measured_heights"][:, :17 * 11].contiguous()
    d["height_noise_offset"], d["u_heights"] = d["height_noise_offset"][:, :187].contiguous(), d["u_heights"][:, :187].contiguous()
    step = foothold.EnvStep(N, DEV, grid=grid, cfg=cfg)
    got = step(max_episode_length=1000, measured_heights=mh, **_fused_kwargs(d, True))
    p = foothold.plan(mh, d["root_states"], d["thigh_pos"], d["commands"], grid=grid)
    reset, tout, mean = foothold.check_termination(d["contact_forces"], d["termination_contact_indices"], d["episode_length_buf"], 1000,
                                                   d["projected_gravity"], d["root_states"], mh, cfg)
    ob = foothold.compute_observations(d["base_ang_vel"], d["projected_gravity"], d["commands"], d["dof_pos"], d["default_dof_pos"],
                                       d["dof_vel"], d["actions"], p["foothold_obs"], d["root_states"], mh, d["forces"],
                                       d["height_noise_offset"], d["u_obs"], d["noise_scale_vec"], d["u_heights"], cfg)
    assert torch.equal(got["optimal_foothold_indice"], p["optimal_foothold_indice"]) and torch.equal(got["reset_buf"].bool(), reset)
    assert torch.equal(got["height_mean"], mean) and torch.equal(got["privileged_obs_buf"], ob["privileged_obs_buf"])
    with pytest.raises(ValueError):
        step(max_episode_length=1000, **_fused_kwargs(d, True))                        # neither heights nor a table
    bad = foothold.EnvStep(N, DEV, cfg=foothold.ObsConfig(num_foothold_obs=4))
    with pytest.raises(_ffi.DtcError):
        bad(max_episode_length=1000, measured_heights=d["measured_heights"], **_fused_kwargs(d, True))
