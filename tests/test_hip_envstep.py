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
