"""Generate the committed golden fixtures by RUNNING THE REFERENCE (read-only import from
/root/reference) on the deterministic synthetic inputs of dtc_amd.synthetic.

    python tests/golden/make_golden.py [gae ppo scorer heights init gru lstm composite observations]

Runs only in the build container (the reference does not exist on the GPU box).  The fixtures
hold outputs only -- inputs are regenerated from seeds on the test side -- so they stay small.
The reference has no tests / golden vectors of its own, hence these files are
the pin for oracle/ (and, through it, for the HIP path).
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "deep-tracking-control_amd"))

import _ref_harness as H  # noqa: E402

H.install()
from dtc_amd import synthetic as S  # noqa: E402
from oracle.ppo_ref import fill_parameters_  # noqa: E402

GOLDEN_THREADS = 1   # fixtures are generated single-threaded (bit-reproducible on one machine)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


def fill_ref_storage(st, data):
    for k, v in data.items():
        if k == "last_values":
            continue
        getattr(st, k).copy_(v)
    st.step = st.num_transitions_per_env


# ------------------------------------------------------------------------------------ G1
def gen_gae():
    from rsl_rl.storage import RolloutStorage
    out = {}
    for N in (64, 4096):
        data = S.rollout(N, 24, seed=4)
        st = RolloutStorage(N, 24, [53], [1389], [265], [12])
        fill_ref_storage(st, data)
        st.compute_returns(data["last_values"], 0.99, 0.95)
        ret, adv = st.returns.squeeze(-1).numpy(), st.advantages.squeeze(-1).numpy()
        stride = 1 if N == 64 else 97
        out[f"returns_{N}"] = ret.reshape(-1)[::stride].copy()
        out[f"advantages_{N}"] = adv.reshape(-1)[::stride].copy()
        out[f"sums_{N}"] = np.array([ret.astype(np.float64).sum(), np.abs(ret).astype(np.float64).sum(),
                                     adv.astype(np.float64).sum(), (adv.astype(np.float64) ** 2).sum()])
    # dones at t=0 and t=T-1, last_values != 0
    N = 16
    data = S.rollout(N, 24, seed=5)
    data["dones"][0, :4] = 1
    data["dones"][23, 2:8] = 1
    lv = torch.linspace(-1, 1, N).unsqueeze(1)
    st = RolloutStorage(N, 24, [53], [1389], [265], [12])
    fill_ref_storage(st, data)
    st.compute_returns(lv, 0.99, 0.95)
    out["returns_edge"] = st.returns.squeeze(-1).numpy()
    out["advantages_edge"] = st.advantages.squeeze(-1).numpy()
    save("gae", **out)


# ------------------------------------------------------------------------------- G2 / G3
def _ref_alg(N, seed_fill=None, **kw):
    from rsl_rl.algorithms import PPO
    from rsl_rl.modules import ActorCriticDecoder
    with H.quiet():
        torch.manual_seed(3)
        ac = ActorCriticDecoder(53, 1389, 12)
        if seed_fill is not None:
            fill_parameters_(ac, seed_fill)
        alg = PPO(ac, device="cpu", learning_rate=1e-3, entropy_coef=0.003, **kw)
        alg.init_storage(N, 24, [53], [1389], [265], [12])
    return alg


def _sample_idx(numel, k=64):
    return (np.arange(k, dtype=np.int64) * 2654435761 % numel).astype(np.int64)


def _run_ref_update(N, seed_fill, max_steps=None, **kw):
    """Run the reference's PPO.update with its random draws replaced by dtc_amd.synthetic's,
    recording per-step scalars from inside the reference's own update() frame."""
    import torch.nn as nn
    alg = _ref_alg(N, seed_fill, **kw)
    data = S.rollout(N, 24, seed=4)
    fill_ref_storage(alg.storage, data)
    alg.storage.compute_returns(data["last_values"], 0.99, 0.95)
    perm, eps1, eps2 = S.update_noise(N, 24, alg.num_mini_batches, alg.num_learning_epochs, seed=123)
    queue = []
    for k in range(eps1.shape[0]):
        queue += [eps1[k], eps2[k]]
    rec = dict(recons=[], vel=[], kld=[], height=[], vae_gnorm=[], surrogate=[], value=[], entropy=[],
               kl_mean=[], lr=[], gnorm=[], w_samples=[])
    names = [k for k, _ in alg.actor_critic.named_parameters()]
    params = dict(alg.actor_critic.named_parameters())
    probe = {k: _sample_idx(params[k].numel(), 8) for k in
             ("vae.terrain_encoder.0.weight", "vae.terrain_decoder.4.weight", "vae.cenet_encoder.0.weight",
              "vae.latent_var.weight", "vae.cenet_decoder.0.weight", "actor_body.0.weight",
              "actor_body.6.weight", "critic_body.0.weight")}

    class Stop(Exception):
        pass

    orig = dict(randperm=torch.randperm, randn_like=torch.randn_like, clip=nn.utils.clip_grad_norm_)
    state = dict(which="vae", steps=0)

    def fake_randperm(n, *a, **k):
        assert n == perm.numel()
        return perm.clone()

    def fake_randn_like(t, *a, **k):
        e = queue.pop(0)
        assert e.shape == t.shape
        return e.clone()

    def fake_clip(parameters, max_norm, *a, **k):
        tn = orig["clip"](parameters, max_norm, *a, **k)
        rec["vae_gnorm" if state["which"] == "vae" else "gnorm"].append(float(tn))
        return tn

    vae_zero, main_zero = alg.vae_optimizer.zero_grad, alg.optimizer.zero_grad
    main_step = alg.optimizer.step

    def vae_zero_grad(*a, **k):
        loc = sys._getframe(1).f_locals
        for key, var in (("recons", "recons_loss"), ("vel", "vel_loss"), ("kld", "kld_loss"),
                         ("height", "height_loss")):
            rec[key].append(float(loc[var]))
        state["which"] = "vae"
        return vae_zero(*a, **k)

    def main_zero_grad(*a, **k):
        loc = sys._getframe(1).f_locals
        rec["surrogate"].append(float(loc["surrogate_loss"]))
        rec["value"].append(float(loc["value_loss"]))
        rec["entropy"].append(float(loc["entropy_batch"].mean()))
        rec["kl_mean"].append(float(loc["kl_mean"]) if "kl_mean" in loc else 0.0)
        rec["lr"].append(float(alg.learning_rate))
        state["which"] = "main"
        return main_zero(*a, **k)

    def main_step_wrapped(*a, **k):
        r = main_step(*a, **k)
        rec["w_samples"].append(np.concatenate(
            [params[n].detach().reshape(-1)[torch.from_numpy(ix)].numpy() for n, ix in probe.items()]))
        state["steps"] += 1
        if max_steps is not None and state["steps"] >= max_steps:
            raise Stop()
        return r

    torch.randperm, torch.randn_like, nn.utils.clip_grad_norm_ = fake_randperm, fake_randn_like, fake_clip
    alg.vae_optimizer.zero_grad, alg.optimizer.zero_grad = vae_zero_grad, main_zero_grad
    alg.optimizer.step = main_step_wrapped
    ret = None
    try:
        torch.manual_seed(123)
        ret = alg.update()
    except Stop:
        pass
    finally:
        torch.randperm, torch.randn_like = orig["randperm"], orig["randn_like"]
        nn.utils.clip_grad_norm_ = orig["clip"]
    out = {k: np.asarray(v, dtype=np.float64) for k, v in rec.items()}
    if ret is not None:
        out["update_return"] = np.asarray([float(x) for x in ret], dtype=np.float64)
    return out, alg


def gen_ppo():
    torch.set_num_threads(GOLDEN_THREADS)
    out = {}
    # G2: forward quantities at filled weights, B = 1536 (64 envs x 24)
    alg = _ref_alg(64, seed_fill=11)
    ac = alg.actor_critic
    data = S.rollout(64, 24, seed=4)
    flat = {k: v.flatten(0, 1) for k, v in data.items() if k != "last_values"}
    g = torch.Generator().manual_seed(99)
    eps = torch.randn(1536, 16, generator=g)
    orig = torch.randn_like
    torch.randn_like = lambda t, *a, **k: eps.clone()
    try:
        with torch.no_grad():
            e = ac.vae.cenet_encoder(flat["observation_histories"])
            lv_raw = ac.vae.latent_var(e).clone()
            mu, lv, z = ac.vae.cenet_forward(flat["observation_histories"])
            l_t = ac.vae.terrain_encoder(flat["privileged_observations"][:, :693])
            ac.update_distribution(flat["observations"], flat["observation_histories"], flat["privileged_observations"])
            mean = ac.action_mean.clone()
            logp = ac.get_actions_log_prob(flat["actions"])
            ent = ac.entropy
            val = ac.evaluate(flat["observations"], flat["privileged_observations"], flat["base_vel"])
    finally:
        torch.randn_like = orig
    out["fwd_n_outliers"] = np.array([(lv != lv_raw).sum().item()])
    out["fwd_median"] = np.array([lv[lv != lv_raw][0].item()])
    out["fwd_mu"] = mu.numpy()[::7]
    out["fwd_lv"] = lv.numpy()[::7]
    out["fwd_z"] = z.numpy()[::7]
    out["fwd_l_t"] = l_t.numpy()[::37, ::5]
    out["fwd_mean"] = mean.numpy()[::7]
    out["fwd_logp"] = logp.numpy()
    out["fwd_entropy"] = ent.numpy()[:4]
    out["fwd_value"] = val.numpy().reshape(-1)
    # G3a: config 1 (64 envs x 24), all 20 steps, free running, adaptive LR, default-style init fill
    r, _ = _run_ref_update(64, seed_fill=11)
    out.update({"u64_" + k: v for k, v in r.items()})
    # G3b: same but 1 epoch x 4 mini-batches with fixed LR (slow error growth -> tight pin)
    r, _ = _run_ref_update(64, seed_fill=11, num_learning_epochs=1, schedule="fixed")
    out.update({"u64f_" + k: v for k, v in r.items()})
    # G3c: config 2 (4096 x 24), first 4 mini-batch steps
    r, _ = _run_ref_update(4096, seed_fill=11, max_steps=4)
    out.update({"u4096_" + k: v for k, v in r.items()})
    save("ppo", **out)


def gen_init():
    """state_dict fingerprint of the reference's own initialisation under torch.manual_seed(3)."""
    alg = _ref_alg(8)
    sd = alg.actor_critic.state_dict()
    keys = list(sd.keys())
    sums = np.array([sd[k].double().sum().item() for k in keys])
    asums = np.array([sd[k].double().abs().sum().item() for k in keys])
    shapes = np.array([list(sd[k].shape) + [0] * (2 - sd[k].dim()) for k in keys])
    save("init", keys=np.array(keys), sums=sums, abs_sums=asums, shapes=shapes)


# ------------------------------------------------------------------------------------ G4
def _ref_scorer(inp):
    from legged_gym.envs.base.legged_robot_dtc import LeggedRobotDTC
    from legged_gym.envs.lite3.lite3_dtc_config import Lite3DTCCfg
    N = inp["root_states"].shape[0]
    noop = lambda *a, **k: None
    m = types.SimpleNamespace()
    m.cfg, m.num_envs, m.device, m.num_bodies = Lite3DTCCfg(), N, "cpu", 17
    m.gym = types.SimpleNamespace(refresh_actor_root_state_tensor=noop, refresh_net_contact_force_tensor=noop,
                                  refresh_rigid_body_state_tensor=noop)
    m.sim = m.viewer = None
    m.enable_viewer_sync = m.debug_viz = False
    m.episode_length_buf = torch.zeros(N, dtype=torch.long)
    m.common_step_counter = 0
    m.root_states = inp["root_states"].clone()
    m.base_quat, m.base_lin_vel, m.base_ang_vel = torch.zeros(N, 4), torch.zeros(N, 3), torch.zeros(N, 3)
    m.base_pos, m.projected_gravity = torch.zeros(N, 3), torch.zeros(N, 3)
    m.gravity_vec = torch.tensor([0., 0., -1.]).repeat(N, 1)
    m.lin_vel_buffer, m.ang_vel_buffer, m.cmd_buffer = torch.zeros(10, N, 2), torch.zeros(10, N, 1), torch.zeros(10, N, 4)
    m.commands = inp["commands"].clone()
    rbs = torch.zeros(N, 17, 13)
    m.thigh_indices, m.feet_indices = torch.tensor([2, 6, 10, 14]), torch.tensor([4, 8, 12, 16])
    rbs[:, m.thigh_indices, 0:3] = inp["thigh_pos"]
    m.rigid_body_state = rbs.view(N * 17, 13)
    m.height_points = S.height_points().unsqueeze(0).repeat(N, 1, 1)
    m.measured_heights = inp["measured_heights"].clone()
    for k in ("_post_physics_step_callback", "check_termination", "compute_reward", "reset_idx",
              "compute_observations"):
        setattr(m, k, noop)
    m.reset_buf = torch.zeros(N, dtype=torch.long)
    m.last_actions_2, m.last_actions, m.actions = torch.zeros(N, 12), torch.zeros(N, 12), torch.zeros(N, 12)
    m.last_dof_vel, m.dof_vel = torch.zeros(N, 12), torch.zeros(N, 12)
    m.last_root_vel, m.last_foot_velocities = torch.zeros(N, 6), torch.zeros(N, 4, 3)
    m.rotate_positions = LeggedRobotDTC.rotate_positions
    LeggedRobotDTC.post_physics_step(m)
    return m


def scorer_edge_inputs():
    """Edge cases: nominal foothold outside the grid, all-exception rows, exact ties, flat."""
    inp = S.scorer_inputs(64, seed=21)
    mh, root = inp["measured_heights"], inp["root_states"]
    mh[0:8] = root[0:8, 2:3] + 2.0                      # every point exceptional -> idx 0
    mh[8:16] = root[8:16, 2:3] - 0.32                   # perfectly flat -> ties between equal scores
    inp["thigh_pos"][16:24, :, :2] += 5.0               # nominal footholds far outside the grid
    mh[24:32, ::2] = root[24:32, 2:3] - 3.0             # half of the points exceptional
    inp["commands"][32:40] = 0.0
    root[40:48, 3:7] = torch.tensor([0., 0., 0., 1.])   # identity attitude
    return inp


from cases import scorer_extra_inputs  # noqa: E402  (shared with the tests; imports nothing of the reference)


def gen_scorer():
    torch.set_num_threads(GOLDEN_THREADS)
    out = {}
    for tag in ("seed2", "bench", "slopes"):
        m = _ref_scorer(scorer_extra_inputs(tag))
        srt = np.sort(m.foothold_score.numpy(), axis=1)
        out[tag + "_idx"] = m.optimal_foothold_indice.squeeze(1).numpy().astype(np.int16)
        out[tag + "_gap"] = (srt[:, 1, :] - srt[:, 0, :]).astype(np.float32)
        out[tag + "_foothold_obs"] = m.foothold_obs.numpy()[::8]
        out[tag + "_pred"] = m.pred_footholds.numpy()[::8]
    for tag, inp in (("main", S.scorer_inputs(8192, seed=7)), ("edge", scorer_edge_inputs())):
        m = _ref_scorer(inp)
        sc = m.foothold_score.numpy()
        srt = np.sort(sc, axis=1)
        out[tag + "_idx"] = m.optimal_foothold_indice.squeeze(1).numpy().astype(np.int16)
        out[tag + "_gap"] = (srt[:, 1, :] - srt[:, 0, :]).astype(np.float32)
        out[tag + "_best"] = srt[:, 0, :].astype(np.float32)
        out[tag + "_nominal_idx"] = m.nominal_footholds_indice.numpy().astype(np.int16)
        out[tag + "_foothold_obs"] = m.foothold_obs.numpy()[::4]
        out[tag + "_world"] = m.optimal_footholds_world.numpy()[::4]
        out[tag + "_pred"] = m.pred_footholds.numpy()[::4]
        out[tag + "_pred_to_robot"] = m.pred_footholds_to_robot.numpy()[::4]
        out[tag + "_slope_sample"] = m.slope.numpy()[::64]
        out[tag + "_score_sample"] = sc[::64]
    # a10 / f3: rewards consuming the planner output (run as unbound methods on a tiny mock)
    inp = S.scorer_inputs(512, seed=7)
    m = _ref_scorer(inp)
    g = torch.Generator().manual_seed(5)
    mock = types.SimpleNamespace(device="cpu", optimal_footholds_world=m.optimal_footholds_world,
                                 foot_positions=m.optimal_footholds_world + 0.05 * torch.randn(512, 4, 3, generator=g),
                                 contact_filt=torch.rand(512, 4, generator=g) < 0.6)
    mock.foot_positions[:, :, 2] = 0.02 * torch.randn(512, 4, generator=g)
    from legged_gym.envs.base.legged_robot_dtc import LeggedRobotDTC
    out["rew_tracking"] = LeggedRobotDTC._reward_tracking_optimal_footholds(mock).numpy()
    out["rew_miss"] = LeggedRobotDTC._reward_foothold_miss(mock).numpy()
    save("scorer", **out)


# ------------------------------------------------------------------------------------ G6
def synthetic_height_table(rows=1760, cols=1120, seed=31):
    g = torch.Generator().manual_seed(seed)
    coarse = torch.randint(-60, 120, (rows // 16, cols // 16), generator=g)
    tab = coarse.repeat_interleave(16, 0).repeat_interleave(16, 1)
    tab = tab + torch.randint(-2, 3, (rows, cols), generator=g)
    return tab.to(torch.int16)


def gen_heights():
    from legged_gym.envs.base.legged_robot import LeggedRobot
    from legged_gym.envs.lite3.lite3_dtc_config import Lite3DTCCfg
    N = 2048
    cfg = Lite3DTCCfg()
    inp = S.scorer_inputs(N, seed=9)
    root = inp["root_states"]
    root[:8, 0] = torch.tensor([-30., -19.99, 0., 67.9, 68.0, 100., 20., 20.])   # clip paths
    root[:8, 1] = torch.tensor([-30., 0., -19.99, 35.9, 36.0, 100., -25., 40.])
    m = types.SimpleNamespace(cfg=cfg, num_envs=N, num_height_points=693, device="cpu",
                              terrain=types.SimpleNamespace(cfg=cfg.terrain),
                              height_samples=synthetic_height_table(),
                              height_points=S.height_points().unsqueeze(0).repeat(N, 1, 1),
                              base_quat=root[:, 3:7].clone(), root_states=root)
    h = LeggedRobot._get_heights(m)
    save("heights", heights=h.numpy()[::4], root_override=root[:8, :2].numpy())


# ------------------------------------------------------------------------------------ G5
def gru_case(N=16, seed=4):
    """Synthetic recurrent rollout: storage fields + hand-fed hidden states (the reference never records them, F2)."""
    data = S.rollout(N, 24, seed=seed)
    data["dones"][:, 0] = 0                                   # one env without resets -> a full-length trajectory
    g = torch.Generator().manual_seed(77)
    hid_a = 0.1 * torch.randn(24, 1, N, 512, generator=g)
    hid_c = 0.1 * torch.randn(24, 1, N, 512, generator=g)
    return data, hid_a, hid_c


def gen_gru():
    torch.set_num_threads(GOLDEN_THREADS)
    from rsl_rl.modules import ActorCriticRecurrent
    from rsl_rl.storage import RolloutStorage
    N = 16
    with H.quiet():
        torch.manual_seed(3)
        ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                                  activation='elu', rnn_type='gru', rnn_hidden_size=512, rnn_num_layers=1)
    fill_parameters_(ac, 21)
    data, hid_a, hid_c = gru_case(N)
    st = RolloutStorage(N, 24, [53], [1389], [265], [12])
    fill_ref_storage(st, data)
    st.saved_hidden_states_a, st.saved_hidden_states_c = [hid_a.clone()], [hid_c.clone()]
    out = dict(keys=np.array(list(ac.state_dict().keys())))
    for i, b in enumerate(st.reccurent_mini_batch_generator(4, 1)):
        (obs_b, cobs_b, act_b, val_b, adv_b, ret_b, lp_b, mu_b, sg_b, (ha, hc), masks) = b
        with torch.no_grad():
            ac.act(obs_b, masks=masks, hidden_states=ha)
            mean = ac.action_mean.clone()
            value = ac.evaluate(cobs_b, masks=masks, hidden_states=hc)
        out[f"mb{i}_shape"] = np.array(list(obs_b.shape) + list(masks.shape) + list(ha.shape))
        out[f"mb{i}_mask_sum"] = np.array([int(masks.sum())])
        out[f"mb{i}_mean"] = mean.numpy()
        out[f"mb{i}_value"] = value.numpy()
        out[f"mb{i}_obs_sum"] = np.array([obs_b.double().sum().item(), cobs_b.double().sum().item()])
    # rollout-mode (inference) forward over 3 consecutive steps
    ac.memory_a.hidden_states = None
    ac.memory_c.hidden_states = None
    seq = []
    with torch.no_grad():
        for t in range(3):
            ac.act(data["observations"][t])
            seq.append(ac.action_mean.clone().numpy())
    out["rollout_means"] = np.stack(seq)
    save("gru", **out)


# ------------------------------------------------------------------------------------ LSTM (the reference's default rnn_type), 2 layers
def lstm_case(N=16, seed=6, H=256, L=2):
    data = S.rollout(N, 24, seed=seed)
    data["dones"][:, 0] = 0
    g = torch.Generator().manual_seed(78)
    mk = lambda: 0.1 * torch.randn(24, L, N, H, generator=g)
    return data, (mk(), mk()), (mk(), mk())


def gen_lstm():
    torch.set_num_threads(GOLDEN_THREADS)
    from rsl_rl.modules import ActorCriticRecurrent
    from rsl_rl.storage import RolloutStorage
    N = 16
    with H.quiet():
        torch.manual_seed(5)
        ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                                  activation='elu', rnn_type='lstm', rnn_hidden_size=256, rnn_num_layers=2)
    fill_parameters_(ac, 23)
    data, hid_a, hid_c = lstm_case(N)
    st = RolloutStorage(N, 24, [53], [1389], [265], [12])
    fill_ref_storage(st, data)
    st.saved_hidden_states_a, st.saved_hidden_states_c = [h.clone() for h in hid_a], [h.clone() for h in hid_c]
    out = dict(keys=np.array(list(ac.state_dict().keys())))
    for i, b in enumerate(st.reccurent_mini_batch_generator(4, 1)):
        (obs_b, cobs_b, act_b, val_b, adv_b, ret_b, lp_b, mu_b, sg_b, (ha, hc), masks) = b
        with torch.no_grad():
            ac.act(obs_b, masks=masks, hidden_states=ha)
            mean = ac.action_mean.clone()
            value = ac.evaluate(cobs_b, masks=masks, hidden_states=hc)
        out[f"mb{i}_shape"] = np.array(list(obs_b.shape) + list(masks.shape) + list(ha[0].shape) + [len(ha)])
        out[f"mb{i}_mean"] = mean.numpy()
        out[f"mb{i}_value"] = value.numpy()
    ac.memory_a.hidden_states = None
    ac.memory_c.hidden_states = None
    seq, vals = [], []
    with torch.no_grad():
        for t in range(3):
            ac.act(data["observations"][t])
            seq.append(ac.action_mean.clone().numpy())
            vals.append(ac.evaluate(data["privileged_observations"][t]).clone().numpy())
    out["rollout_means"], out["rollout_values"] = np.stack(seq), np.stack(vals)
    save("lstm", **out)


# ------------------------------------------------------------------------------------ f4: deployment path + checkpoint layout
def gen_teacher():
    """`ActorCriticDecoder.act_teacher` (actor_critic_decoder.py:504-538, reached through `act_expert` /
    `get_inference_policy(env_t=True)`) at filled weights, and the layout of the checkpoint dictionary written by
    `OnPolicyRunner.save` (on_policy_runner.py:249-255)."""
    torch.set_num_threads(GOLDEN_THREADS)
    alg = _ref_alg(64, seed_fill=11)
    ac = alg.actor_critic
    d = S.rollout(64, 24, seed=4)
    obs, hist, priv = (d[k].flatten(0, 1)[:512] for k in ("observations", "observation_histories", "privileged_observations"))
    with torch.no_grad():
        mean = ac.act_expert(dict(obs=obs, obs_history=hist, privileged_obs=priv))
    ckpt = {'model_state_dict': ac.state_dict(), 'optimizer_state_dict': alg.optimizer.state_dict(), 'iter': 7, 'infos': None}
    osd = ckpt['optimizer_state_dict']
    save("teacher", mean=mean.numpy(), ckpt_keys=np.array(list(ckpt.keys())),
         model_keys=np.array(list(ckpt['model_state_dict'].keys())),
         model_shapes=np.array([str(tuple(v.shape)) for v in ckpt['model_state_dict'].values()]),
         opt_keys=np.array(list(osd.keys())), group_keys=np.array(sorted(osd['param_groups'][0].keys())),
         n_group_params=np.array([len(osd['param_groups'][0]['params'])]))


def gen_student():
    """`ActorCriticDecoder.act_student` (actor_critic_decoder.py:459-502) and `.adapt_bootstrap_probability` (:404-407).
    As written, `act_student` stops at its first statement: it reads `self.cenet_encoder`, `self.latent_mu` (as a layer),
    `self.actor_student` and `self.exporter`, none of which the class defines (the encoder and the head live under `self.vae`, the only
    actor is `actor_body`, and the exporter loads a file from an absolute path of the authors' machine).  The vector here is the body of the
    reference's own function run with exactly those names bound: cenet_encoder / latent_mu -> the `vae` modules of the same name,
    actor_student -> actor_body (same input width: obs 53 + 16 + 3 + 512), count = 1 and a no-op exporter (its output is discarded)."""
    torch.set_num_threads(GOLDEN_THREADS)
    alg = _ref_alg(64, seed_fill=11)
    ac = alg.actor_critic
    ac.cenet_encoder, ac.latent_mu, ac.actor_student = ac.vae.cenet_encoder, ac.vae.latent_mu, ac.actor_body
    ac.count, ac.exporter = 1, (lambda x: None)
    d = S.rollout(64, 24, seed=4)
    obs, hist, priv = (d[k].flatten(0, 1)[:512] for k in ("observations", "observation_histories", "privileged_observations"))
    g = torch.Generator().manual_seed(91)
    lidar = torch.randn(512, 512, generator=g)
    with torch.no_grad():
        mean = ac.act_student(obs, hist, priv, lidar)
    rew = [torch.rand(4096, generator=g) + 0.5, torch.randn(4096, generator=g) * 3 + 1.0, torch.full((64,), 2.0),
           torch.randn(1024, generator=g) * 0.01 - 1.0]
    probs = np.array([type(ac).adapt_bootstrap_probability(ac, r) for r in rew])
    save("student", mean=mean.numpy(), lidar_seed=np.array([91]), bootstrap_prob=probs,
         **{f"rew{i}": r.numpy() for i, r in enumerate(rew)})


# ------------------------------------------------------------------------------------ f3: observations / termination
def gen_observations():
    """`LeggedRobotDTC.compute_observations` and `.check_termination` (legged_robot_dtc.py:229-288) run as
    unbound methods on a mock env built from dtc_amd.synthetic.env_state; the two torch.rand_like draws are
    replaced by the generator's uniforms."""
    from legged_gym.envs.base.legged_robot_dtc import LeggedRobotDTC
    from legged_gym.envs.lite3.lite3_dtc_config import Lite3DTCCfg
    N = 1024
    s = S.env_state(N, seed=13)
    cfg = Lite3DTCCfg()
    m = types.SimpleNamespace(cfg=cfg, device="cpu", num_envs=N)
    m.obs_scales = cfg.normalization.obs_scales
    m.commands_scale = torch.tensor([m.obs_scales.lin_vel, m.obs_scales.lin_vel, m.obs_scales.ang_vel])
    m.cpg_phase_information = None
    for k in ("base_ang_vel", "projected_gravity", "commands", "dof_pos", "dof_vel", "actions", "foothold_obs",
              "root_states", "measured_heights", "forces", "height_noise_offset", "noise_scale_vec", "contact_forces",
              "episode_length_buf", "termination_contact_indices"):
        setattr(m, k, s[k].clone())
    m.default_dof_pos = s["default_dof_pos"].unsqueeze(0)
    m.add_noise = True
    m.max_episode_length = 1000
    draws = [s["u_heights"], s["u_obs"]]              # order of the calls at :278 and :287
    orig = torch.rand_like
    torch.rand_like = lambda t, *a, **k: draws.pop(0).clone()
    try:
        LeggedRobotDTC.compute_observations(m)
    finally:
        torch.rand_like = orig
    assert not draws
    LeggedRobotDTC.check_termination(m)
    mean = torch.mean(m.root_states[:, 2].unsqueeze(1) - m.measured_heights[:, 210:483].clip(min=-0.), dim=1)
    save("observations", obs=m.obs_buf.numpy(), priv_sample=m.privileged_obs_buf.numpy()[::32],
         priv_sum=m.privileged_obs_buf.double().sum(dim=1).numpy(), heights_sample=m.heights.numpy()[::32],
         reset=m.reset_buf.numpy().astype(np.uint8), time_out=m.time_out_buf.numpy().astype(np.uint8),
         height_mean=mean.numpy(), base_height_target=np.array([cfg.rewards.base_height_target]))


# ------------------------------------------------------------------------------------ config 5 composite
def composite_case(N=16, seed=4):
    data, hid_a, hid_c = gru_case(N, seed)
    g = torch.Generator().manual_seed(78)
    eps = torch.randn(4, 24 * (N // 4), 16, generator=g)
    G1 = torch.randn(24 * (N // 4), 12, generator=g)          # cotangents of the probe functional
    G2 = torch.randn(24 * (N // 4), 1, generator=g)
    return data, hid_a, hid_c, eps, G1, G2


def gen_composite():
    """BASELINE config 5 (build-defined): imported `Vae` feature builders feeding the imported
    `ActorCriticRecurrent(584, 752, 12, gru 512)` over the imported pad / un-pad helpers.  Outputs: action mean /
    value / log-prob / entropy per recurrent mini-batch and the parameter gradients of the probe functional
    sum(mean*G1) + sum(value*G2) (autograd through the imported modules: GRU BPTT -> features -> encoders)."""
    torch.set_num_threads(GOLDEN_THREADS)
    import torch.nn as nn
    from rsl_rl.modules import ActorCriticRecurrent
    from rsl_rl.modules.actor_critic_decoder import Vae
    from rsl_rl.utils import split_and_pad_trajectories, unpad_trajectories
    N, nmb = 16, 4

    class Comp(nn.Module):
        def __init__(self):
            super().__init__()
            self.vae = Vae()
            self.acr = ActorCriticRecurrent(584, 752, 12, actor_hidden_dims=[512, 256, 128],
                                            critic_hidden_dims=[512, 256, 128], activation='elu', rnn_type='gru',
                                            rnn_hidden_size=512, rnn_num_layers=1)

    with H.quiet():
        torch.manual_seed(3)
        m = Comp()
    fill_parameters_(m, 23)
    data, hid_a, hid_c, eps_all, G1, G2 = composite_case(N)
    T, mb = 24, N // nmb
    dones = data["dones"]
    lwd = torch.zeros(T, N, dtype=torch.bool)
    lwd[1:] = dones[:-1, :, 0].bool()
    lwd[0] = True
    out = dict(keys=np.array([k.replace("acr.", "") for k in m.state_dict().keys()]))
    randn_like = torch.randn_like
    for i in range(nmb):
        a, b = i * mb, (i + 1) * mb
        sl = lambda k: data[k][:, a:b].flatten(0, 1)
        obs, hist, priv, bv = sl("observations"), sl("observation_histories"), sl("privileged_observations"), sl("base_vel")
        eps = eps_all[i]
        torch.randn_like = lambda t: eps                       # reparameterize (actor_critic_decoder.py:283) draws here
        try:
            mu, lv, z = m.vae.cenet_forward(hist)
        finally:
            torch.randn_like = randn_like
        l_t = m.vae.terrain_encoder(priv[:, :693])
        fa = torch.cat((obs, z, mu[:, :3], l_t), dim=-1).view(T, mb, -1)
        fc = torch.cat((obs, bv, priv[:, 693:696], priv[:, 696:]), dim=-1).view(T, mb, -1)
        d = dones[:, a:b]
        pa, masks = split_and_pad_trajectories(fa, d)
        pc, _ = split_and_pad_trajectories(fc, d)
        pick = lambda h: h[:, :, a:b].permute(2, 0, 1, 3)[lwd[:, a:b].permute(1, 0)].transpose(1, 0).contiguous()
        actions = m.acr.act(pa, masks=masks, hidden_states=pick(hid_a))
        mean = m.acr.action_mean
        value = m.acr.evaluate(pc, masks=masks, hidden_states=pick(hid_c))
        logp = m.acr.get_actions_log_prob(data["actions"][:, a:b])
        ent = m.acr.entropy
        m.zero_grad()
        ((mean.flatten(0, 1) * G1).sum() + (value.flatten(0, 1) * G2).sum()).backward()
        out[f"mb{i}_mean"] = mean.detach().flatten(0, 1).numpy()
        out[f"mb{i}_value"] = value.detach().flatten(0, 1).numpy()
        out[f"mb{i}_logp"] = logp.detach().flatten(0, 1).numpy()
        out[f"mb{i}_entropy"] = ent.detach().flatten(0, 1).numpy()
        out[f"mb{i}_ntraj"] = np.array([masks.shape[1]])
        if i == 0:
            for k, p in m.named_parameters():
                if p.grad is not None:
                    gname = k.replace("acr.", "")
                    out["g_" + gname] = p.grad.numpy().copy() if p.grad.numel() <= 4096 else \
                        np.concatenate([p.grad.flatten()[_sample_idx(p.grad.numel())].numpy(),
                                        np.array([p.grad.double().sum().item(), p.grad.double().pow(2).sum().item()])]).astype(np.float64)
    save("composite", **out)


TASKS = dict(lstm=gen_lstm, teacher=gen_teacher, student=gen_student, observations=gen_observations, composite=gen_composite, gru=gen_gru, gae=gen_gae, ppo=gen_ppo, init=gen_init, scorer=gen_scorer, heights=gen_heights)

if __name__ == "__main__":
    for t in (sys.argv[1:] or list(TASKS)):
        TASKS[t]()
