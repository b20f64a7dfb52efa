"""BASELINE config 5: GRU + CE-net + foothold-obs composite model (build-defined).
CPU: oracle/composite_ref.py against tests/golden/composite.npz -- the same composition built from the IMPORTED
reference classes (Vae, ActorCriticRecurrent, split_and_pad_trajectories, unpad_trajectories): forward outputs of
every recurrent mini-batch and the parameter gradients of a probe functional (BPTT -> features -> encoders).
GPU: dtc_amd ActorCriticDecoderRecurrent + RecurrentDecoderPPO against the oracle (teacher-forced mini-batch steps)."""
import numpy as np
import pytest
import torch

from dtc_amd import synthetic as S
from oracle import composite_ref as CR
from oracle import ppo_ref as OP

DEV = "cuda:0"
N, NMB, T = 16, 4, 24


def composite_case(seed=4, n=N):
    data = S.rollout(n, T, seed=seed)
    data["dones"][:, 0] = 0
    g = torch.Generator().manual_seed(77)
    hid_a = 0.1 * torch.randn(T, 1, n, 512, generator=g)
    hid_c = 0.1 * torch.randn(T, 1, n, 512, generator=g)
    g = torch.Generator().manual_seed(78)
    eps = torch.randn(4, T * (n // NMB), 16, generator=g)
    G1 = torch.randn(T * (n // NMB), 12, generator=g)
    G2 = torch.randn(T * (n // NMB), 1, generator=g)
    return data, hid_a, hid_c, eps, G1, G2


def oracle_model():
    torch.manual_seed(3)
    return OP.fill_parameters_(CR.RefCompositeAC(), 23)


def oracle_alg(data, n=N, **kw):
    alg = CR.RefCompositePPO(oracle_model(), learning_rate=1e-3, entropy_coef=0.003, **kw)
    alg.init_storage(n, T)
    st = alg.storage
    for k, v in data.items():
        if k != "last_values":
            getattr(st, k).copy_(v)
    st.compute_returns(data["last_values"], 0.99, 0.95)
    return alg


def _sample_idx(numel, k=64):
    return (np.arange(k, dtype=np.int64) * 2654435761 % numel).astype(np.int64)


def test_oracle_matches_reference_composition(golden):
    g = golden("composite")
    data, hid_a, hid_c, eps, G1, G2 = composite_case()
    alg = oracle_alg(data)
    ac = alg.actor_critic
    assert [k.replace("acr.", "") for k in ac.state_dict().keys()] == [str(k) for k in g["keys"]]
    for i, bt in enumerate(CR.recurrent_slices(alg.storage, hid_a, hid_c, NMB)):
        mean, value = alg.forward(bt, eps[i])
        assert bt["hid_a"].shape[1] == int(g[f"mb{i}_ntraj"][0])
        np.testing.assert_allclose(mean.detach().numpy(), g[f"mb{i}_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(value.detach().numpy(), g[f"mb{i}_value"], rtol=1e-5, atol=1e-6)
        dist = torch.distributions.Normal(mean, mean * 0. + ac.std)
        actions = alg.storage.actions.flatten(0, 1)[bt["idx"]]
        np.testing.assert_allclose(dist.log_prob(actions).sum(-1).detach().numpy(), g[f"mb{i}_logp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(dist.entropy().sum(-1).detach().numpy(), g[f"mb{i}_entropy"], rtol=1e-6, atol=1e-6)
        if i == 0:
            ac.zero_grad()
            ((mean * G1).sum() + (value * G2).sum()).backward()
            n_checked = 0
            for k, p in ac.named_parameters():
                k = k.replace("acr.", "")
                if p.grad is None:
                    assert "g_" + k not in g.files, k
                    continue
                ref = g["g_" + k]
                if p.grad.numel() <= 4096:
                    np.testing.assert_allclose(p.grad.numpy(), ref, rtol=2e-4, atol=2e-5 * max(1.0, float(np.abs(ref).max())), err_msg=k)
                else:
                    got = p.grad.flatten()[_sample_idx(p.grad.numel())].numpy()
                    scale = max(1.0, float(np.abs(ref[:64]).max()))
                    np.testing.assert_allclose(got, ref[:64], rtol=2e-4, atol=2e-5 * scale, err_msg=k)
                    np.testing.assert_allclose(p.grad.double().sum().item(), ref[64], rtol=1e-4, atol=1e-3 * scale, err_msg=k)
                    np.testing.assert_allclose(p.grad.double().pow(2).sum().item(), ref[65], rtol=1e-4, err_msg=k)
                n_checked += 1
            # every trainable block takes part: encoders (through the GRU input), both GRUs, both MLPs
            assert n_checked >= 30


# ------------------------------------------------------------------------------------------ GPU (HIP path)
def _strip(sd):
    return {k.replace("acr.", ""): v for k, v in sd.items()}


def _hip_alg(ref, data, n=N, **kw):
    from dtc_amd.algorithms import RecurrentDecoderPPO
    from dtc_amd.modules import ActorCriticDecoderRecurrent
    ac = ActorCriticDecoderRecurrent(53, 1389, 12)
    alg = RecurrentDecoderPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV, **kw)
    alg.init_storage(n, T, [53], [1389], [265], [12])
    ac.load_state_dict(_strip(ref.actor_critic.state_dict()))
    for k, v in data.items():
        if k != "last_values":
            getattr(alg.storage, k).copy_(v.to(DEV))
    alg.storage.compute_returns(data["last_values"].to(DEV), 0.99, 0.95)
    return alg


def _grad_report(grads_ref, alg, which, tol, skip=(), knife_edges=None):
    """Every parameter gradient against the oracle's: max element error relative to the tensor's max and relative L2,
    both <= tol.  knife_edges = (n, B) (full-size mini-batches): n ReLU outputs are masked differently by the two
    (correct) fp32 forwards -- pre-activations that are 0 within rounding, about one (sample, unit) pair per 4e5.  Each
    moves every gradient upstream of it by that sample's share, so the bound becomes the one of
    test_hip_ppo._compare_grads: 99th percentile of the element errors and L2 / 5 within tol + 3 n / B."""
    arena = alg.actor_critic.arena
    worst = []
    if knife_edges is not None:
        tol = tol + 3.0 * knife_edges[0] / knife_edges[1]
    for name, g_ref in grads_ref.items():
        name = name.replace("acr.", "")
        if name.startswith(skip):
            continue
        g = arena.view(alg.captured[which], name).cpu()
        scale = float(g_ref.abs().max()) + 1e-30
        e = ((g - g_ref).abs() / scale).reshape(-1)
        l2 = float((g - g_ref).norm() / (g_ref.norm() + 1e-30))
        if knife_edges is not None:
            q99 = float(torch.quantile(e, 0.99)) if e.numel() > 100 else float(e.max())
            worst.append((max(q99, l2 / 5), name, f"q99={q99:.1e} l2={l2:.1e} max={float(e.max()):.1e}"))
        else:
            worst.append((max(float(e.max()), l2), name, ""))
    worst.sort(reverse=True)
    assert worst[0][0] <= tol, (tol, worst[:6])
    return len(worst)


@pytest.mark.gpu
def test_hip_rollout_mode_forward_carries_state():
    data, hid_a, hid_c, eps, _, _ = composite_case()
    ref = oracle_alg(data)
    alg = _hip_alg(ref, data)
    rac, ac = ref.actor_critic, alg.actor_critic
    ha = hc = None
    g = torch.Generator().manual_seed(5)
    for t in range(3):
        e = torch.randn(N, 16, generator=g)
        obs, hist, priv, bv = (data[k][t] for k in ("observations", "observation_histories", "privileged_observations", "base_vel"))
        with torch.no_grad():
            out, ha = rac.memory_a.rnn(rac.actor_features(obs, hist, priv, e).unsqueeze(0), ha)
            mean_ref = rac.actor(out.squeeze(0))
            out, hc = rac.memory_c.rnn(rac.critic_features(obs, priv, bv).unsqueeze(0), hc)
            val_ref = rac.critic(out.squeeze(0))
        ac.update_distribution(obs.to(DEV), hist.to(DEV), priv.to(DEV), eps=e.to(DEV))
        val = ac.evaluate(obs.to(DEV), priv.to(DEV), bv.to(DEV))
        np.testing.assert_allclose(ac.action_mean.cpu().numpy(), mean_ref.numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(val.cpu().numpy(), val_ref.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ac.memory_a.hidden_states.cpu().numpy(), ha.numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kw,amax_check", [(dict(), False), (dict(schedule="fixed", use_clipped_value_loss=False), False), (dict(), True)])
def test_hip_teacher_forced_minibatches_vs_oracle(kw, amax_check, monkeypatch):
    """Each of the 4 recurrent mini-batches, both optimisation steps, from identical weights and fresh Adam states:
    per-step scalars to 1e-5 rel, every parameter gradient (BPTT through both GRUs, through the feature blocks into
    the CE-net / terrain encoders) to 2e-5 of the tensor's max.  `amax_check`: the same run with DTC_AMAX_CHECK's re-derivation of every
    amax record the recurrent step's converting kernels are handed (the operand-image chain of the VAE step keeps no records)."""
    from dtc_amd import ops
    from dtc_amd.algorithms import ppo as P
    monkeypatch.setattr(ops, "AMAX_CHECK", bool(amax_check))
    data, hid_a, hid_c, eps, _, _ = composite_case()
    g = torch.Generator().manual_seed(79)
    eps2 = torch.randn(4, T * (N // NMB), 16, generator=g)
    for i in range(NMB):
        ref = oracle_alg(data, **kw)
        ref.capture_grads = True
        alg = _hip_alg(ref, data, **kw)
        alg.capture_grads = True
        # the backward pass's data-dependent branches (ReLU signs, CE-net outlier set + median element) follow the oracle's forward
        # (test_hip_ppo._force_oracle_signs): the CE-net encoder gradients are ALWAYS compared, never skipped on a median knife edge
        from test_hip_ppo import _force_oracle_signs
        forced = _force_oracle_signs(ref, alg)
        bt_ref = list(CR.recurrent_slices(ref.storage, hid_a, hid_c, NMB))[i]
        bt = list(alg.recurrent_slices(hid_a.to(DEV), hid_c.to(DEV)))[i]
        assert bt["R"] == bt_ref["hid_a"].shape[1] and torch.equal(bt["idx"].cpu(), bt_ref["idx"])
        rec = OP.StepRecord()
        # VAE step
        ref.vae_step(bt_ref["idx"], eps[i], rec)
        stats = alg.step_minibatch(bt, eps[i].to(DEV), eps2[i].to(DEV), which="vae")
        row = stats.cpu()
        for key, col in (("recons", P.S_RECONS), ("vel", P.S_VEL), ("kld", P.S_KLD), ("height", P.S_HEIGHT),
                         ("vae_gnorm", P.S_VAE_GNORM)):
            assert abs(float(row[col]) - getattr(rec, key)) <= 1e-5 * max(1.0, abs(getattr(rec, key))), (i, key)
        fw = alg.actor_critic._fwd_ws(bt["idx"].numel())
        same_median = int(fw.info[0]) == ref.actor_critic.vae.last_outliers and \
            int(fw.info[1]) == ref.actor_critic.vae.last_median_index
        assert same_median and (not alg.relu_masks or forced[-1][0] == "vae"), (i, forced, fw.info[:2].tolist())
        assert _grad_report(rec.extra["vae_grads"], alg, "vae", 2e-5, ()) >= 20
        # policy step (BPTT) from the oracle's post-VAE-step weights
        alg.actor_critic.load_state_dict(_strip(ref.actor_critic.state_dict()))
        ref.ppo_step(bt_ref, eps2[i], rec)
        stats = alg.step_minibatch(bt, eps[i].to(DEV), eps2[i].to(DEV), which="ppo")
        row = stats.cpu()
        keys = [("surrogate", P.S_SURR), ("value", P.S_VALUE), ("entropy", P.S_ENTROPY), ("gnorm", P.S_GNORM)]
        if ref.schedule == "adaptive":
            keys.append(("kl_mean", P.S_KL))
        for key, col in keys:
            assert abs(float(row[col]) - getattr(rec, key)) <= 1e-5 * max(1.0, abs(getattr(rec, key))), (i, key, float(row[col]), getattr(rec, key))
        assert abs(float(alg.optimizer.lr_dev.item()) - ref.learning_rate) <= 1e-12
        same_median = int(fw.info[0]) == ref.actor_critic.vae.last_outliers and \
            int(fw.info[1]) == ref.actor_critic.vae.last_median_index
        assert same_median and (not alg.relu_masks or forced[-1][0] == "ppo"), (i, forced, fw.info[:2].tolist())
        n = _grad_report(rec.extra["grads"], alg, "main", 2e-5, ())
        assert n >= 35            # std, 2 MLPs, 2 GRUs, CE-net encoder + heads, terrain encoder


@pytest.mark.gpu
def test_hip_teacher_forced_minibatch_full_size():
    """BASELINE configs[4]'s model at 4096 envs per GPU: the first recurrent mini-batch (1024 envs x 24
    steps, ~1500 padded trajectories), both optimisation steps, against the CPU oracle -- scalars 1e-5 rel, every
    parameter gradient 5e-5 of its max (BPTT through both GRUs into the feature blocks and the encoders)."""
    from dtc_amd.algorithms import ppo as P
    n = 4096
    data, hid_a, hid_c, eps, _, _ = composite_case(n=n)
    g = torch.Generator().manual_seed(79)
    eps2 = torch.randn(1, T * (n // NMB), 16, generator=g)
    ref = oracle_alg(data, n)
    ref.capture_grads = True
    alg = _hip_alg(ref, data, n)
    alg.capture_grads = True
    bt_ref = next(iter(CR.recurrent_slices(ref.storage, hid_a, hid_c, NMB)))
    bt = next(iter(alg.recurrent_slices(hid_a.to(DEV), hid_c.to(DEV))))
    assert bt["R"] == bt_ref["hid_a"].shape[1] and bt["R"] > 1200 and torch.equal(bt["idx"].cpu(), bt_ref["idx"])
    rec = OP.StepRecord()
    # the backward pass's two data-dependent branches (ReLU signs, CE-net outlier set + median element) follow the oracle's
    # forward: fp32 knife edges stay out of the gradient comparison (test_hip_ppo._force_oracle_signs)
    from test_hip_ppo import _force_oracle_signs
    forced = _force_oracle_signs(ref, alg)
    ref.vae_step(bt_ref["idx"], eps[0], rec)
    row = alg.step_minibatch(bt, eps[0].to(DEV), eps2[0].to(DEV), which="vae").cpu()
    print("full-size composite, VAE step: HIP forward's own (outliers, median element, entries classified differently):", alg.own_branch,
          "oracle:", (ref.actor_critic.vae.last_outliers, ref.actor_critic.vae.last_median_index))
    for key, col in (("recons", P.S_RECONS), ("vel", P.S_VEL), ("kld", P.S_KLD), ("height", P.S_HEIGHT), ("vae_gnorm", P.S_VAE_GNORM)):
        assert abs(float(row[col]) - getattr(rec, key)) <= 1e-5 * max(1.0, abs(getattr(rec, key))), (key, float(row[col]), getattr(rec, key))
    fw = alg.actor_critic._fwd_ws(bt["idx"].numel())
    same_median = int(fw.info[0]) == ref.actor_critic.vae.last_outliers and int(fw.info[1]) == ref.actor_critic.vae.last_median_index
    if not same_median:                                      # (forced above: cannot happen while the sign records are on)
        pytest.fail("full-size composite, VAE step: the CE-net encoder gradients would be skipped (median landed on another element)")
    skip = ()
    from test_hip_ppo import _relu_mask_mismatches
    edges = _relu_mask_mismatches(ref, alg, "vae")
    print("  ReLU knife edges:", edges[0])
    assert _grad_report(rec.extra["vae_grads"], alg, "vae", 2e-5, skip, knife_edges=edges) >= 20
    alg.actor_critic.load_state_dict(_strip(ref.actor_critic.state_dict()))
    ref.ppo_step(bt_ref, eps2[0], rec)
    row = alg.step_minibatch(bt, eps[0].to(DEV), eps2[0].to(DEV), which="ppo").cpu()
    assert not alg.relu_masks or [w for w, _ in forced] == ["vae", "ppo"], forced
    print("full-size composite, policy step: HIP forward's own (outliers, median element, entries classified differently):", alg.own_branch,
          "oracle:", (ref.actor_critic.vae.last_outliers, ref.actor_critic.vae.last_median_index))
    for key, col in (("surrogate", P.S_SURR), ("value", P.S_VALUE), ("entropy", P.S_ENTROPY), ("gnorm", P.S_GNORM), ("kl_mean", P.S_KL)):
        assert abs(float(row[col]) - getattr(rec, key)) <= 1e-5 * max(1.0, abs(getattr(rec, key))), (key, float(row[col]), getattr(rec, key))
    assert abs(float(alg.optimizer.lr_dev.item()) - ref.learning_rate) <= 1e-12
    same_median = int(fw.info[0]) == ref.actor_critic.vae.last_outliers and int(fw.info[1]) == ref.actor_critic.vae.last_median_index
    if not same_median:
        pytest.fail("full-size composite, policy step: the CE-net encoder gradients would be skipped (median landed on another element)")
    skip = ()
    edges = _relu_mask_mismatches(ref, alg, "ppo")
    print("  ReLU knife edges:", edges[0])
    assert _grad_report(rec.extra["grads"], alg, "main", 2e-5, skip, knife_edges=edges) >= 35


@pytest.mark.gpu
def test_hip_rollout_and_update_end_to_end():
    """act / process_env_step record the hidden states, update() consumes them; 2 updates stay finite."""
    from dtc_amd.algorithms import RecurrentDecoderPPO
    from dtc_amd.modules import ActorCriticDecoderRecurrent
    torch.manual_seed(3)
    n = 64
    ac = ActorCriticDecoderRecurrent(53, 1389, 12)
    alg = RecurrentDecoderPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
    alg.init_storage(n, T, [53], [1389], [265], [12])
    d = S.rollout(n, T, seed=9, device=DEV)
    for it in range(2):
        for t in range(T):
            alg.act(d["observations"][t], d["privileged_observations"][t], d["observation_histories"][t], d["base_vel"][t])
            alg.process_env_step(d["rewards"][t, :, 0], d["dones"][t, :, 0], d["next_observations"][t], {})
        assert alg.storage.saved_hidden_states_a[0].shape == (T, 1, n, 512)
        if it == 0:
            assert float(alg.storage.saved_hidden_states_a[0][0].abs().max()) == 0.0       # first step starts from zeros
            assert float(alg.storage.saved_hidden_states_a[0][5].abs().max()) > 0.0
        alg.compute_returns(d["observations"][-1], d["privileged_observations"][-1], d["base_vel"][-1])
        out = alg.update()
        assert all(np.isfinite(out)), out
    assert 1e-5 <= alg.learning_rate <= 1e-2


@pytest.mark.gpu
def test_runner_drives_the_composite_by_name():
    """OnPolicyRunner resolves `ActorCriticDecoderRecurrent` / `RecurrentDecoderPPO` from train_cfg and runs the same
    rollout / learn loop as for the reference classes (on_policy_runner.py:86-190)."""
    from dtc_amd.env import ReplayEnv
    from dtc_amd.runners import OnPolicyRunner
    cfg = dict(runner=dict(policy_class_name="ActorCriticDecoderRecurrent", algorithm_class_name="RecurrentDecoderPPO",
                           num_steps_per_env=24, save_interval=10), algorithm=dict(learning_rate=1e-3), policy=dict())
    r = OnPolicyRunner(ReplayEnv(32, DEV), cfg, log_dir=None, device=DEV)
    r.learn(2)
    assert r.current_learning_iteration == 2
    sd = r.alg.actor_critic.state_dict()
    assert all(torch.isfinite(v).all() for v in sd.values())
    assert r.alg.storage.saved_hidden_states_a[0].shape == (24, 1, 32, 512)


def _by_name(opt_sd, ref_module, hip_module):
    """torch Adam state (indexed by the position of the parameter in the ORACLE module's parameter list) re-indexed for the HIP module,
    whose parameters are registered in another order (std first): matched by name."""
    ref_pos = {k.replace("acr.", ""): i for i, (k, _) in enumerate(ref_module.named_parameters())}
    state = {}
    for j, (name, _) in enumerate(hip_module.named_parameters()):
        i = ref_pos[name]
        if i in opt_sd["state"]:
            state[j] = opt_sd["state"][i]
    return dict(state=state, param_groups=opt_sd["param_groups"])


@pytest.mark.gpu
def test_two_consecutive_updates_vs_oracle():
    """RecurrentDecoderPPO.update() twice on two DIFFERENT rollouts (1 epoch x 4 recurrent mini-batches each) against the oracle stepping
    the same mini-batches.  Each update starts from the oracle's weights / both Adam states / learning rate, so the first mini-batch of
    EVERY update is a 1e-5 comparison: nothing the trainer keeps between updates (packed rollout rows and their generation keys, amax
    records, padded workspaces) may leak into the second one (the bug class of 61f564b).  Later steps run free inside the F4 envelope."""
    from dtc_amd.algorithms import ppo as P
    n = 64
    kw = dict(num_learning_epochs=1)
    cols = dict(recons=P.S_RECONS, vel=P.S_VEL, kld=P.S_KLD, height=P.S_HEIGHT, vae_gnorm=P.S_VAE_GNORM,
                surrogate=P.S_SURR, value=P.S_VALUE, entropy=P.S_ENTROPY, kl_mean=P.S_KL, gnorm=P.S_GNORM)
    envelope = (1e-5, 3e-4, 1.5e-3, 5e-3)
    ref = alg = None
    worst = []
    for u, seed in enumerate((4, 12)):
        data, hid_a, hid_c, _, _, _ = composite_case(seed=seed, n=n)
        g = torch.Generator().manual_seed(300 + u)
        B = T * (n // NMB)
        e1, e2 = torch.randn(NMB, B, 16, generator=g), torch.randn(NMB, B, 16, generator=g)
        if ref is None:
            ref = oracle_alg(data, n, **kw)
            alg = _hip_alg(ref, data, n, **kw)
            assert sorted(alg.actor_critic.state_dict().keys()) == sorted(_strip(ref.actor_critic.state_dict()).keys())
        else:
            for side, to in ((ref.storage, lambda v: v), (alg.storage, lambda v: v.to(DEV))):
                for k, v in data.items():
                    if k != "last_values":
                        getattr(side, k).copy_(to(v))
                side.compute_returns(to(data["last_values"]), 0.99, 0.95)
        alg.storage.step = T
        alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid_a.to(DEV)], [hid_c.to(DEV)]
        alg.actor_critic.load_state_dict(_strip(ref.actor_critic.state_dict()))
        alg.optimizer.load_state_dict(_by_name(ref.optimizer.state_dict(), ref.actor_critic, alg.actor_critic))
        alg.vae_optimizer.load_state_dict(_by_name(ref.vae_optimizer.state_dict(), ref.actor_critic.vae, alg.actor_critic.vae))
        alg.learning_rate = ref.learning_rate
        alg.vae_optimizer.set_lr(5e-4)
        recs = [ref.step(bt, e1[i], e2[i]) for i, bt in enumerate(CR.recurrent_slices(ref.storage, hid_a, hid_c, NMB))]
        alg.update(e1.to(DEV), e2.to(DEV))
        rows = alg.last_update_stats
        assert rows.shape[0] == NMB
        for k, rec in enumerate(recs):
            for key, c in cols.items():
                refv = getattr(rec, key)
                worst.append((abs(float(rows[k, c]) - refv) / max(1.0, abs(refv)) / envelope[k], u, k, key, float(rows[k, c]), refv))
        assert abs(alg.learning_rate - ref.learning_rate) <= 1e-12, (u, alg.learning_rate, ref.learning_rate)
    first_steps = [w for w in worst if w[2] == 0]
    assert max(first_steps)[0] <= 1.0, sorted(first_steps, reverse=True)[:4]          # update 2, step 0 included: 1e-5
    assert max(worst)[0] <= 1.0, sorted(worst, reverse=True)[:6]
