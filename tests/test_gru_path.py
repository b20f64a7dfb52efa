"""Recurrent (GRU) actor-critic path, BASELINE config 3.
CPU: oracle/gru_ref.py against the fixture captured from the reference's own ActorCriticRecurrent / Memory /
split_and_pad / reccurent_mini_batch_generator (tests/golden/gru.npz).
GPU: dtc_amd ActorCriticRecurrent + RecurrentPPO against the oracle (forward, every parameter gradient of the
BPTT step, scalars, learning rate)."""
import numpy as np
import pytest
import torch

from dtc_amd import synthetic as S
from oracle import gru_ref as GR
from oracle import ppo_ref as OP

DEV = "cuda:0"
N = 16


def gru_case(seed=4, n=N):
    data = S.rollout(n, 24, seed=seed)
    data["dones"][:, 0] = 0
    g = torch.Generator().manual_seed(77)
    hid_a = 0.1 * torch.randn(24, 1, n, 512, generator=g)
    hid_c = 0.1 * torch.randn(24, 1, n, 512, generator=g)
    return data, hid_a, hid_c


def oracle_model():
    torch.manual_seed(3)
    return OP.fill_parameters_(GR.RefActorCriticRecurrent(), 21)


def oracle_storage(data, n=N):
    st = OP.RefStorage(n, 24)
    for k, v in data.items():
        if k != "last_values":
            getattr(st, k).copy_(v)
    st.compute_returns(data["last_values"], 0.99, 0.95)
    return st


def test_oracle_matches_reference_modules(golden):
    g = golden("gru")
    ac = oracle_model()
    assert list(ac.state_dict().keys()) == [str(k) for k in g["keys"]]
    data, hid_a, hid_c = gru_case()
    st = oracle_storage(data)
    alg = GR.RefRecurrentPPO(ac)
    for i, b in enumerate(GR.recurrent_batches(st, hid_a, hid_c, 4)):
        shape = g[f"mb{i}_shape"]
        assert list(b["obs"].shape) == list(shape[:3]) and list(b["masks"].shape) == list(shape[3:5])
        assert list(b["hid_a"].shape) == list(shape[5:8]) and int(b["masks"].sum()) == int(g[f"mb{i}_mask_sum"][0])
        sums = g[f"mb{i}_obs_sum"]
        assert abs(b["obs"].double().sum().item() - sums[0]) < 1e-6 and abs(b["cobs"].double().sum().item() - sums[1]) < 1e-5
        with torch.no_grad():
            mean, value = alg.forward(b)
        np.testing.assert_allclose(mean.numpy(), g[f"mb{i}_mean"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(value.numpy(), g[f"mb{i}_value"], rtol=1e-5, atol=1e-6)
    # rollout mode: state carried across steps
    h = None
    with torch.no_grad():
        for t in range(3):
            out, h = ac.memory_a.rnn(data["observations"][t].unsqueeze(0), h)
            np.testing.assert_allclose(ac.actor(out.squeeze(0)).numpy(), g["rollout_means"][t], rtol=1e-5, atol=1e-6)


def _oracle64(ref, st, batch, **kw):
    """The oracle's step once more in float64 (same weights, same optimiser state, same batch): the reference the PER-ROW gradient bounds
    below are measured against -- the float32 oracle's own rows are up to 4.5e-6 of their row maximum away from it."""
    import copy

    def to64(o):
        if torch.is_tensor(o):
            return o.double() if o.is_floating_point() else o
        if isinstance(o, dict):
            return {k: to64(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return type(o)(to64(v) for v in o)
        return o
    r64 = GR.RefRecurrentPPO(copy.deepcopy(ref.ac).double(), learning_rate=ref.learning_rate, entropy_coef=0.003, **kw)
    r64.capture_grads = True
    st64 = copy.copy(st)
    for k, v in vars(st).items():
        if torch.is_tensor(v) and v.is_floating_point():
            setattr(st64, k, v.double())
    return r64.step(st64, to64(batch))["grads"]


def _row_err(g, g64):
    """largest |g - g64| relative to the maximum of its ROW (one output feature of a weight matrix; vectors: the tensor's maximum)"""
    g, g64 = g.double(), g64.double()
    if g64.dim() == 2:
        return float(((g - g64).abs() / g64.abs().amax(1, keepdim=True).clamp_min(1e-300)).max())
    return float((g - g64).abs().max() / g64.abs().max().clamp_min(1e-300))


ROW_TOL = 2e-5          # per-row bound of the BPTT gradients against the float64 oracle (measured: see the tests' prints under -s)


def _hip_pair(n=N, **kw):
    from dtc_amd.algorithms import RecurrentPPO
    from dtc_amd.modules import ActorCriticRecurrent
    ref_ac = oracle_model()
    ref = GR.RefRecurrentPPO(ref_ac, learning_rate=1e-3, entropy_coef=0.003, **kw)
    ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                              activation='elu', rnn_type='gru', rnn_hidden_size=512, rnn_num_layers=1)
    alg = RecurrentPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV, **kw)
    alg.init_storage(n, 24, [53], [1389], [12])
    ac.load_state_dict(ref_ac.state_dict())
    data, hid_a, hid_c = gru_case(n=n)
    st = oracle_storage(data, n)
    for k, v in data.items():
        if k not in ("last_values", "observation_histories"):      # the recurrent storage keeps no obs history
            getattr(alg.storage, k).copy_(v.to(DEV))
    alg.storage.compute_returns(data["last_values"].to(DEV), 0.99, 0.95)
    alg.storage.saved_hidden_states_a = [hid_a.to(DEV)]
    alg.storage.saved_hidden_states_c = [hid_c.to(DEV)]
    return ref, alg, st, hid_a, hid_c


@pytest.mark.gpu
def test_state_dict_and_rollout_forward():
    ref, alg, st, _, _ = _hip_pair()
    ac = alg.actor_critic
    assert list(ac.state_dict().keys()) == list(ref.ac.state_dict().keys())
    h = None
    for t in range(3):
        obs = st.observations[t]
        with torch.no_grad():
            out, h = ref.ac.memory_a.rnn(obs.unsqueeze(0), h)
            exp = ref.ac.actor(out.squeeze(0))
        got = ac.act_inference(obs.to(DEV))
        np.testing.assert_allclose(got.cpu().numpy(), exp.numpy(), rtol=1e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(schedule="fixed", use_clipped_value_loss=False)])
def test_recurrent_minibatch_step_vs_oracle(kw):
    """All four mini-batches of one epoch, teacher-forced: forward, scalars, LR, every parameter gradient (BPTT)."""
    from dtc_amd.algorithms import ppo as P
    ref, alg, st, hid_a, hid_c = _hip_pair(**kw)
    ref.capture_grads = alg.capture_grads = True
    gen = alg.storage.reccurent_mini_batch_generator(4, 1)
    for i, (b_ref, b_hip) in enumerate(zip(GR.recurrent_batches(st, hid_a, hid_c, 4), gen)):
        alg.actor_critic.load_state_dict(ref.ac.state_dict())
        alg.optimizer.load_state_dict(ref.optimizer.state_dict())
        alg.learning_rate = ref.learning_rate
        assert torch.equal(b_hip[0].cpu(), b_ref["obs"]) and torch.equal(b_hip[10].cpu(), b_ref["masks"])
        g64 = _oracle64(ref, st, b_ref, **kw)               # (before ref.step: the same weights and optimiser state)
        rec = ref.step(st, b_ref)
        row = alg.step_minibatch(b_hip, i * 4, (i + 1) * 4).cpu()
        ac = alg.actor_critic
        np.testing.assert_allclose(ac._actor_outs[-1].cpu().numpy().reshape(24, 4, 12), rec["mean"].numpy(), rtol=1e-5, atol=2e-6)
        np.testing.assert_allclose(ac._critic_outs[-1].cpu().numpy().reshape(24, 4, 1), rec["value_out"].numpy(), rtol=1e-5, atol=2e-6)
        for key, col in (("surrogate", P.S_SURR), ("value", P.S_VALUE), ("entropy", P.S_ENTROPY), ("gnorm", P.S_GNORM)):
            assert abs(float(row[col]) - rec[key]) <= 1e-5 * max(1.0, abs(rec[key])), (i, key, float(row[col]), rec[key])
        if "kl_mean" in rec:
            assert abs(float(row[P.S_KL]) - rec["kl_mean"]) <= 1e-5 * max(1.0, abs(rec["kl_mean"]))
        assert float(alg.optimizer.lr_dev.item()) == rec["lr"]
        for name, g_ref in rec["grads"].items():
            g = ac.arena.view(alg.captured["main"], name).cpu()
            scale = float(g_ref.abs().max()) + 1e-30
            err = float((g - g_ref).abs().max()) / scale
            assert err <= 5e-5, (i, name, err, scale)
        rows = {name: _row_err(ac.arena.view(alg.captured["main"], name).cpu(), g) for name, g in g64.items()}
        worst = max(rows, key=rows.get)
        print(f"mini-batch {i}: worst per-row gradient error vs the float64 oracle {rows[worst]:.2e} ({worst}); "
              f"float32 oracle {max(_row_err(rec['grads'][n], g) for n, g in g64.items()):.2e}")
        assert rows[worst] <= ROW_TOL, (i, worst, rows[worst])
        assert len(rec["grads"]) == 25


@pytest.mark.gpu
def test_recurrent_minibatch_step_full_size():
    """BASELINE configs[2] at its real size: 4096 envs x 24 steps, one recurrent mini-batch of 1024 envs
    (~1500 padded trajectories) teacher-forced against the CPU oracle.  At this size the HIP path runs what the 16-env
    cases never reach: the split-precision kernels on every wide layer (128 x 128 tiles), the GRU time steps on csrc/gru_s3.hip
    (fused forward step at R ~ 1500 rows, the six-chunk recurrent data gradient), the input projection and the W_ih / W_hh weight
    gradients over the valid rows of the padded batch only, grouped weight gradients with 8-24 batch slices."""
    from dtc_amd.algorithms import ppo as P
    n = 4096
    ref, alg, st, hid_a, hid_c = _hip_pair(n=n)
    ref.capture_grads = alg.capture_grads = True
    b_ref = next(iter(GR.recurrent_batches(st, hid_a, hid_c, 4)))
    b_hip = next(iter(alg.storage.reccurent_mini_batch_generator(4, 1)))
    assert torch.equal(b_hip[10].cpu(), b_ref["masks"]) and b_ref["masks"].shape[1] > 1200
    g64 = _oracle64(ref, st, b_ref)
    rec = ref.step(st, b_ref)
    row = alg.step_minibatch(b_hip, 0, n // 4).cpu()
    ac = alg.actor_critic
    np.testing.assert_allclose(ac._actor_outs[-1].cpu().numpy().reshape(24, n // 4, 12), rec["mean"].numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ac._critic_outs[-1].cpu().numpy().reshape(24, n // 4, 1), rec["value_out"].numpy(), rtol=1e-5, atol=2e-6)
    for key, col in (("surrogate", P.S_SURR), ("value", P.S_VALUE), ("entropy", P.S_ENTROPY), ("gnorm", P.S_GNORM), ("kl_mean", P.S_KL)):
        assert abs(float(row[col]) - rec[key]) <= 1e-5 * max(1.0, abs(rec[key])), (key, float(row[col]), rec[key])
    assert float(alg.optimizer.lr_dev.item()) == rec["lr"]
    for name, g_ref in rec["grads"].items():
        g = ac.arena.view(alg.captured["main"], name).cpu()
        scale = float(g_ref.abs().max()) + 1e-30
        err = float((g - g_ref).abs().max()) / scale
        assert err <= 5e-5, (name, err, scale)
    rows = {name: _row_err(ac.arena.view(alg.captured["main"], name).cpu(), g) for name, g in g64.items()}
    worst = max(rows, key=rows.get)
    print(f"full size: worst per-row gradient error vs the float64 oracle {rows[worst]:.2e} ({worst}); "
          f"float32 oracle {max(_row_err(rec['grads'][n], g) for n, g in g64.items()):.2e}")
    assert rows[worst] <= ROW_TOL, (worst, rows[worst])
    assert len(rec["grads"]) == 25


@pytest.mark.gpu
def test_recurrent_update_and_rollout_api():
    from dtc_amd.algorithms import RecurrentPPO
    from dtc_amd.modules import ActorCriticRecurrent
    torch.manual_seed(0)
    ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                              rnn_type='gru', rnn_hidden_size=512)
    alg = RecurrentPPO(ac, device=DEV, learning_rate=1e-3)
    alg.init_storage(32, 24, [53], [1389], [12])
    g = torch.Generator(device=DEV).manual_seed(1)
    for t in range(24):
        obs = torch.randn(32, 53, device=DEV, generator=g)
        cobs = torch.randn(32, 1389, device=DEV, generator=g)
        alg.act(obs, cobs)
        dones = (torch.rand(32, device=DEV, generator=g) < 0.05)
        alg.process_env_step(0.1 * torch.randn(32, device=DEV, generator=g), dones, {})
    alg.compute_returns(torch.randn(32, 1389, device=DEV, generator=g))
    v, s = alg.update()
    assert np.isfinite(v) and np.isfinite(s) and alg.storage.step == 0
    assert alg.last_update_stats.shape[0] == 20


@pytest.mark.gpu
def test_recurrent_overlapped_schedule_equals_serial_bitwise():
    """RecurrentPPO / RecurrentDecoderPPO: critic recurrence on the second stream + weight-gradient stream vs the
    single-stream schedule, bit-identical weights after a full update (256 envs x 24)."""
    from dtc_amd.algorithms import RecurrentDecoderPPO, RecurrentPPO
    from dtc_amd.modules import ActorCriticDecoderRecurrent, ActorCriticRecurrent
    n = 256
    d = S.rollout(n, 24, seed=9, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(5)
    hid = [0.1 * torch.randn(24, 1, n, 512, generator=g, device=DEV) for _ in range(2)]
    eps = [torch.randn(20, 24 * n // 4, 16, generator=g, device=DEV) for _ in range(2)]
    for kind in ("gru", "composite"):
        out = []
        for overlap in (True, False):
            torch.manual_seed(3)
            if kind == "gru":
                ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                                          activation='elu', rnn_type='gru', rnn_hidden_size=512, rnn_num_layers=1)
                alg = RecurrentPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
                alg.init_storage(n, 24, [53], [1389], [12])
                alg.overlap = overlap
            else:
                ac = ActorCriticDecoderRecurrent(53, 1389, 12)
                alg = RecurrentDecoderPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
                alg.init_storage(n, 24, [53], [1389], [265], [12])
                alg.overlap_wgrad = alg.overlap_lanes = overlap
            for k, v in d.items():
                if k != "last_values" and not (kind == "gru" and k == "observation_histories"):
                    getattr(alg.storage, k).copy_(v)
            alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
            alg.storage.step = 24
            alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid[0]], [hid[1]]
            if kind == "gru":
                alg.update()
            else:
                alg.update(eps[0], eps[1])
            out.append({k: v.clone() for k, v in ac.state_dict().items()})
        for k, v in out[0].items():
            assert torch.equal(v, out[1][k]), (kind, k)


@pytest.mark.gpu
def test_runner_drives_the_recurrent_actor_critic_by_name():
    """BASELINE configs[2] from train_cfg, as the reference's runner resolves classes by name (on_policy_runner.py:38-42,
    60, 67): `ActorCriticRecurrent` with the GRU of the config, trained by `RecurrentPPO` (the upstream PPO step; the
    fork's own `PPO` fails on this model with the same AttributeError as the reference, ppo.py:79)."""
    from dtc_amd.env import ReplayEnv
    from dtc_amd.runners import OnPolicyRunner
    pol = dict(actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128], activation='elu', rnn_type='gru',
               rnn_hidden_size=512, rnn_num_layers=1)
    cfg = dict(runner=dict(policy_class_name="ActorCriticRecurrent", algorithm_class_name="RecurrentPPO",
                           num_steps_per_env=24, save_interval=10), algorithm=dict(learning_rate=1e-3), policy=pol)
    r = OnPolicyRunner(ReplayEnv(32, DEV), cfg, log_dir=None, device=DEV)
    r.learn(2)
    assert r.current_learning_iteration == 2
    assert all(torch.isfinite(v).all() for v in r.alg.actor_critic.state_dict().values())
    assert r.alg.storage.saved_hidden_states_a[0].shape == (24, 1, 32, 512)
    assert r.get_inference_policy()(torch.zeros(32, 53, device=DEV)).shape == (32, 12)
    with pytest.raises(AttributeError, match="vae"):
        OnPolicyRunner(ReplayEnv(32, DEV), dict(cfg, runner=dict(cfg["runner"], algorithm_class_name="PPO")), device=DEV)
    with pytest.raises(AttributeError, match="vae"):       # the MLP actor-critic resolves by name too; only PPO-with-VAE trains here
        OnPolicyRunner(ReplayEnv(32, DEV), dict(cfg, policy=dict(), runner=dict(cfg["runner"], policy_class_name="ActorCritic",
                                                                               algorithm_class_name="PPO")), device=DEV)


@pytest.mark.gpu
def test_recurrent_operand_image_path_next_to_the_converting_kernels():
    """RecurrentPPO at 256 envs x 24 (mini-batch = 1536 rows = 12 row tiles): the step on operand images (input projection from the
    packed valid rows, MLPs on images, W_ih / W_hh / MLP weight gradients as ONE grouped image launch per recurrence with dgh_all
    taken from dtc_gru_bwd's workspace) against the same step on round 4's converting kernels (use_images = False) -- same inputs,
    same weights: forward outputs to 1e-5, every parameter gradient to 2e-5 of its tensor's largest element."""
    from dtc_amd.algorithms import RecurrentPPO
    from dtc_amd.modules import ActorCriticRecurrent
    n = 256
    d = S.rollout(n, 24, seed=11, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(6)
    hid = [0.1 * torch.randn(24, 1, n, 512, generator=g, device=DEV) for _ in range(2)]
    res = []
    for images in (True, False):
        torch.manual_seed(3)
        ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                                  activation='elu', rnn_type='gru', rnn_hidden_size=512, rnn_num_layers=1)
        alg = RecurrentPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
        alg.init_storage(n, 24, [53], [1389], [12])
        alg.use_images, alg.capture_grads = images, True
        for k, v in d.items():
            if k not in ("last_values", "observation_histories"):
                getattr(alg.storage, k).copy_(v)
        alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
        alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid[0]], [hid[1]]
        batch = next(iter(alg.storage.reccurent_mini_batch_generator(4, 1)))
        assert alg._image_mode(24 * n // 4) == images
        row = alg.step_minibatch(batch, 0, n // 4).cpu()
        res.append((ac._actor_outs[-1].clone(), ac._critic_outs[-1].clone(), row, alg.captured["main"].clone(), ac.arena))
    (m1, v1, r1, g1, ar), (m0, v0, r0, g0, _) = res
    np.testing.assert_allclose(m1.cpu().numpy(), m0.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(v1.cpu().numpy(), v0.cpu().numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(r1.numpy(), r0.numpy(), rtol=1e-5, atol=1e-6)
    for name, (off, cnt, _shape) in ar.offsets.items():
        a, b = g1[off:off + cnt], g0[off:off + cnt]
        scale = float(b.abs().max()) + 1e-30
        assert float((a - b).abs().max()) / scale <= 2e-5, (name, float((a - b).abs().max()) / scale)


@pytest.mark.gpu
def test_second_update_packs_the_new_rollout():
    """RecurrentPPO's image path packs the valid rows of the padded observations once per update and mini-batch slot.  The generation
    that keys those images must change from update to update: with the storage refilled between two updates, the second update has to
    read the NEW observations (regression: the key was 1 in every update, so every update after the first reused the first one's
    packed images).  Checked on what the updates READ: every image `_packed_obs` hands out equals a fresh pack of the current
    rollout's valid rows, in both of two consecutive updates on different rollouts."""
    from dtc_amd import h2i
    from dtc_amd.algorithms import RecurrentPPO
    from dtc_amd.modules import ActorCriticRecurrent
    from dtc_amd._ffi import seg, segmat
    n = 64
    torch.manual_seed(3)
    ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                              activation='elu', rnn_type='gru', rnn_hidden_size=512, rnn_num_layers=1)
    alg = RecurrentPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
    alg.init_storage(n, 24, [53], [1389], [12])
    g = torch.Generator(device=DEV).manual_seed(5)
    hid = [0.1 * torch.randn(24, 1, n, 512, generator=g, device=DEV) for _ in range(2)]
    seen = {}
    for seed in (9, 10):
        d = S.rollout(n, 24, seed=seed, device=DEV)
        for k, v in d.items():
            if k not in ("last_values", "observation_histories"):
                getattr(alg.storage, k).copy_(v)
        alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
        alg.storage.step = 24
        alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid[0]], [hid[1]]
        packed = []
        orig = alg._packed_obs

        def spy(name, x, unpad_idx, M, dev, orig=orig, packed=packed):
            im = orig(name, x, unpad_idx, M, dev)
            x2 = x.float().contiguous().view(-1, x.shape[-1])
            fresh = h2i.HImage(M, x.shape[-1], dev).pack(segmat([seg(x2, 0, x2.shape[1], gather=True)], unpad_idx), M)
            packed.append(bool(torch.equal(im.buf, fresh.buf)))
            return im
        alg._packed_obs = spy
        alg.update()
        alg._packed_obs = orig
        seen[seed] = packed
    assert len(seen[9]) == len(seen[10]) == 40 and all(seen[9])
    assert all(seen[10]), f"{seen[10].count(False)} of 40 packed observation images of the second update are stale"


@pytest.mark.gpu
def test_two_consecutive_updates_vs_oracle():
    """RecurrentPPO.update() twice on two DIFFERENT rollouts against the oracle stepping the same eight mini-batches (1 epoch x 4 per
    update).  Both sides start each update from the oracle's weights / Adam state / learning rate, so the first step of EVERY update is a
    1e-5 comparison -- whatever the trainer carries from update to update (packed observation images, their generation keys, workspaces)
    must not leak into the second one (the bug class of 61f564b: the second update trained on the first update's packed observations).
    Later steps of an update run free: Adam turns rounding noise of near-zero gradients into +-lr flips, so the bound widens per step."""
    from dtc_amd.algorithms import ppo as P
    from dtc_amd.algorithms import RecurrentPPO
    from dtc_amd.modules import ActorCriticRecurrent
    n = 64
    ref_ac = oracle_model()
    ref = GR.RefRecurrentPPO(ref_ac, learning_rate=1e-3, entropy_coef=0.003)
    ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                              activation='elu', rnn_type='gru', rnn_hidden_size=512, rnn_num_layers=1)
    alg = RecurrentPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV, num_learning_epochs=1)
    alg.init_storage(n, 24, [53], [1389], [12])
    envelope = (1e-5, 3e-4, 1.5e-3, 5e-3)
    worst = []
    for u, seed in enumerate((4, 12)):
        data, hid_a, hid_c = gru_case(seed=seed, n=n)
        st = oracle_storage(data, n)
        for k, v in data.items():
            if k not in ("last_values", "observation_histories"):
                getattr(alg.storage, k).copy_(v.to(DEV))
        alg.storage.compute_returns(data["last_values"].to(DEV), 0.99, 0.95)
        alg.storage.step = 24
        alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid_a.to(DEV)], [hid_c.to(DEV)]
        ac.load_state_dict(ref.ac.state_dict())
        alg.optimizer.load_state_dict(ref.optimizer.state_dict())
        alg.learning_rate = ref.learning_rate
        recs = [ref.step(st, b) for b in GR.recurrent_batches(st, hid_a, hid_c, 4)]
        alg.update()
        rows = alg.last_update_stats
        assert rows.shape[0] == 4
        for k, rec in enumerate(recs):
            for key, col in (("surrogate", P.S_SURR), ("value", P.S_VALUE), ("entropy", P.S_ENTROPY), ("gnorm", P.S_GNORM), ("kl_mean", P.S_KL)):
                err = abs(float(rows[k, col]) - rec[key]) / max(1.0, abs(rec[key]))
                worst.append((err / envelope[k], u, k, key, float(rows[k, col]), rec[key]))
        assert abs(alg.learning_rate - ref.learning_rate) <= 1e-12, (u, alg.learning_rate, ref.learning_rate)
    first_steps = [w for w in worst if w[2] == 0]
    assert max(first_steps)[0] <= 1.0, sorted(first_steps, reverse=True)[:4]          # update 2, step 0 included: 1e-5
    assert max(worst)[0] <= 1.0, sorted(worst, reverse=True)[:6]


@pytest.mark.gpu
def test_recurrent_minibatch_step_vs_oracle_on_the_persistent_recurrence():
    """The teacher-forced mini-batch steps once more with the opt-in persistent forward recurrence (csrc/gru_seq.hip: ONE launch for all 24
    time steps, W_hh slices resident in LDS as two-term fp16) in place of the per-step launches: same bounds against the oracle."""
    from dtc_amd import _ffi
    lib = _ffi.lib()
    lib.dtc_set_gru_seq(1)
    try:
        assert lib.dtc_gru_seq_supported(24, 20, 512, 0) == 1
        test_recurrent_minibatch_step_vs_oracle(dict())
        assert lib.dtc_gru_seq_status(1) == 0
    finally:
        lib.dtc_set_gru_seq(-1)
