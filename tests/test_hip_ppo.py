"""PPO.update on the HIP path vs the CPU oracle (oracle/ppo_ref.py), teacher-forced per mini-batch
step as SURVEY.md F4 prescribes: before every step the HIP model, both Adam states and the learning
rate are synchronised from the oracle, both run ONE mini-batch on identical indices / noise, and the
per-step scalars, gradient norms, every parameter gradient and the updated weights are compared.
Each mini-batch is two optimisation steps (VAE, PPO); both start from the oracle's exact state.
Tolerance: 1e-5 * max(1, |x|) on scalars (BASELINE.json north_star); every parameter gradient: max(99th percentile of the
element errors, L2 error / 5) <= 2e-5 of the tensor's max (+ 3/B per ReLU knife edge); weights: 2e-6 abs on every element
when the HIP optimiser steps on the ORACLE's gradient (the real arena, the batch's real sizes), a sanity bound on the
weights from its own gradients (see _compare_weights).
GPU only."""
import numpy as np
import pytest
import torch

from dtc_amd import synthetic as S

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-5
# Largest single-element error of any parameter gradient, relative to the tensor's max, that a teacher-forced step may show
# next to the q99 / L2 criterion of _compare_grads (which it complements: q99 and L2 cannot see one wrong element of 3.5e5).
# Measured over the five full-size (B = 24576) teacher-forced steps: 6.6e-6 at most (vae.latent_mu.weight, policy step).
MAX_ELEM_TOL = 2e-5


def _pair(N, seed=4, **kw):
    """(oracle RefPPO on CPU, HIP PPO on GPU) with identical filled weights and rollout."""
    from dtc_amd.algorithms import PPO
    from dtc_amd.modules import ActorCriticDecoder
    from oracle import ppo_ref as OP
    torch.manual_seed(3)
    ref_ac = OP.fill_parameters_(OP.RefActorCriticDecoder(), 11)
    ref = OP.RefPPO(ref_ac, learning_rate=1e-3, entropy_coef=0.003, **kw)
    ref.init_storage(N, 24)
    torch.manual_seed(3)
    ac = ActorCriticDecoder(53, 1389, 12)
    alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV, **kw)
    alg.init_storage(N, 24, [53], [1389], [265], [12])
    ac.load_state_dict(ref_ac.state_dict())
    d = S.rollout(N, 24, seed=seed)
    for k, v in d.items():
        if k == "last_values":
            continue
        getattr(ref.storage, k).copy_(v)
        getattr(alg.storage, k).copy_(v.to(DEV))
    ref.storage.compute_returns(d["last_values"], 0.99, 0.95)
    alg.storage.compute_returns(d["last_values"].to(DEV), 0.99, 0.95)
    return ref, alg


def _sync_from_oracle(ref, alg):
    alg.actor_critic.load_state_dict(ref.actor_critic.state_dict())
    alg.optimizer.load_state_dict(ref.optimizer.state_dict())
    alg.vae_optimizer.load_state_dict(ref.vae_optimizer.state_dict())
    alg.learning_rate = ref.learning_rate
    alg.vae_optimizer.set_lr(5e-4)


def _close(a, b, tol=TOL):
    return abs(a - b) <= tol * max(1.0, abs(b))


def _compare_scalars(k, rec, row, ref, keys):
    from dtc_amd.algorithms import ppo as P
    cols = dict(recons=P.S_RECONS, vel=P.S_VEL, kld=P.S_KLD, height=P.S_HEIGHT, vae_gnorm=P.S_VAE_GNORM,
                surrogate=P.S_SURR, value=P.S_VALUE, entropy=P.S_ENTROPY, kl_mean=P.S_KL, gnorm=P.S_GNORM)
    for key in keys:
        if key == "kl_mean" and ref.schedule != "adaptive":
            continue                      # the reference only evaluates the KL under the adaptive schedule
        got = float(row[cols[key]])
        assert _close(got, getattr(rec, key)), (k, key, got, getattr(rec, key))


def _relu_mask_mismatches(ref, alg, which):
    """Number of ReLU outputs whose sign differs between the oracle's and the HIP forward pass (fp32
    knife edges: a pre-activation that is 0 within rounding)."""
    B = next(iter(ref.relu_masks.values())).shape[0]
    fw, tw = alg.actor_critic._fwd_ws(B), alg._train_ws(B)
    # (value(): the wide hidden activations may exist as activation images only -- decoded here)
    mine = {"cenet_encoder.1": fw.value("e1"), "terrain_encoder.1": fw.value("t1"), "terrain_encoder.3": fw.value("t2")}
    if which == "vae":
        mine.update({"cenet_decoder.1": tw.value("c1"), "cenet_decoder.3": tw.value("c2"), "terrain_decoder.1": tw.value("d1"),
                     "terrain_decoder.3": tw.value("d2")})
    return sum(int(((buf.cpu() > 0) != ref.relu_masks[name]).sum()) for name, buf in mine.items()), B


_MASK_NAMES = {"cenet_encoder.1": "e1", "terrain_encoder.1": "t1", "terrain_encoder.3": "t2", "cenet_decoder.1": "c1",
               "cenet_decoder.3": "c2", "terrain_decoder.1": "d1", "terrain_decoder.3": "d2"}


def _pack_sign_record(pos):
    """bool [B, W] -> the int16 sign-record words of dtc_linear_fwd_mask (include/dtc_hip.h)."""
    B, W = pos.shape
    bits = pos.view(B // 32, 4, 2, 4, W).permute(0, 2, 1, 3, 4).reshape(B // 32, 2, 16, W).to(torch.int32)
    words = (bits << torch.arange(16, dtype=torch.int32).view(1, 1, 16, 1)).sum(dim=2).reshape(-1, W)
    return torch.from_numpy(words.numpy().astype(np.uint16).view(np.int16))


def _force_oracle_signs(ref, alg):
    """Teacher-force the two data-dependent branches of the backward pass with the ORACLE's decisions, between the HIP forward
    and backward pass of a step:
    * the ReLU derivative pattern (sign records).  A pre-activation that is 0 within fp32 rounding lands on different sides
      in two correct forwards (about one (sample, unit) pair per 4e5) and moves every upstream gradient by that sample's share;
    * the CE-net outlier classification and the median element (actor_critic_decoder.py:293-299): every replaced entry sends
      its gradient to the ONE median element, and at B = 24576 the non-outliers around the median are ~2e-6 apart, so GEMM
      rounding noise decides which element that is (~8 % of the calls differ).  Entries classified differently also get the
      oracle's log-variance, so the backward differentiates the function the oracle differentiated.
    With both forced the gradient comparison measures the backward kernels, not the knife edges.  What the HIP forward
    decided on its own is kept in alg.own_branch = (outliers, median index, entries classified differently) and reported.
    Returns the list of forced steps."""
    forced = []

    def hook(fw, which):
        torch.cuda.synchronize()
        n = 0
        for ref_name, name in _MASK_NAMES.items():
            buf = fw._masks.get(name)
            if buf is None or ref_name not in ref.relu_masks or (which == "ppo" and name in ("c1", "c2", "d1", "d2")):
                continue
            words = _pack_sign_record(ref.relu_masks[ref_name])
            assert words.numel() == buf.numel(), (name, words.shape, buf.shape)
            buf.copy_(words.reshape(-1).to(buf.device))
            n += 1
        vae = ref.actor_critic.vae
        want = vae.last_outlier_mask.to(fw.mask.device)
        differ = fw.mask.bool() != want
        alg.own_branch = (int(fw.info[0]), int(fw.info[1]), int(differ.sum()))
        if alg.own_branch[2]:
            fw.mulv[:, 19:][differ] = vae.last_logvar.to(fw.mulv.device)[differ]
        fw.mask.copy_(want.to(torch.uint8))
        fw.info[:2] = torch.tensor([vae.last_outliers, vae.last_median_index], dtype=torch.int32, device=fw.info.device)
        forced.append((which, n))
        torch.cuda.synchronize()

    alg.own_branch = None
    alg.after_forward_hook = hook if alg.relu_masks else None
    return forced


def _compare_grads(k, which, grads_ref, ref, alg, n_expected, strict=False, forced=False):
    """Pre-clip gradient of every parameter (manual backward through the HIP kernels vs torch autograd):
    99 % of the elements of every tensor within `tol` of the tensor's max, whole tensor within tol in L2.
    tol = 2e-5 when every ReLU of the step is masked identically by both implementations.  A ReLU
    pre-activation that is 0 within fp32 rounding (about one (sample, unit) pair per 4e5) is legitimately
    masked differently by two correct fp32 forwards and moves every upstream gradient by that sample's
    share, so each such knife edge widens the bound by 3/B (tools/debug_step2.py demonstrates one)."""
    arena = alg.actor_critic.arena
    assert len(grads_ref) == n_expected
    n_mis, B = _relu_mask_mismatches(ref, alg, which)
    # CE-net outlier rule: all replaced entries send their gradient to the ONE median element
    # (actor_critic_decoder.py:293-299).  If a borderline entry is classified differently (thresholds from
    # fp64 vs torch's fp32 mean/std; ~8 % of the calls at B = 24576) the non-outlier count changes parity
    # and the lower median moves to the neighbouring order statistic -- another element -- so the
    # concentrated gradient lands elsewhere: the CE-net encoder gradients are then not comparable.
    fw = alg.actor_critic._fwd_ws(B)
    vae = ref.actor_critic.vae
    # ... and at B = 24576 the 3.7e5 non-outliers are ~2e-6 apart around the median, so fp32 GEMM noise
    # (3e-7) can even swap WHICH element is the median: compare the element index, not just the value.
    same_median = int(fw.info[0]) == vae.last_outliers and int(fw.info[1]) == vae.last_median_index
    own = getattr(alg, "own_branch", None) if alg.after_forward_hook is not None else None
    BRANCH_LOG.append((B, which, bool(same_median)))          # full-size tests ASSERT that the compared branch ran (below)
    if B >= 4096:
        note = "" if own is None else (f" (teacher-forced; the HIP forward's own choice: {own[0]} outliers, median element {own[1]}, "
                                       f"{own[2]} entries classified differently from the oracle's {vae.last_outliers} / {vae.last_median_index})")
        print(f"[step {k} {which}] B={B}: CE-net encoder gradients {'compared' if same_median else 'SKIPPED (median on another element)'}"
              f"{note}; ReLU knife edges {n_mis}")
    if strict:
        assert n_mis == 0 and same_median, (n_mis, int(fw.info[0]), vae.last_outliers,
                                            float(fw.info[2:3].view(torch.float32)), vae.last_median)
    # records teacher-forced (every ReLU layer of this batch size has one): the flat bound; otherwise 3/B per knife edge
    tol = 2e-5 if forced else 2e-5 + 3.0 * n_mis / B
    report = []
    for name, g_ref in grads_ref.items():
        if not same_median and name.startswith(("vae.cenet_encoder", "vae.latent_")):
            continue
        g = arena.view(alg.captured[which], name).cpu()
        scale = float(g_ref.abs().max()) + 1e-30
        err = ((g - g_ref).abs() / scale).reshape(-1)
        q99 = float(torch.quantile(err, 0.99)) if err.numel() > 100 else float(err.max())
        l2 = float((g - g_ref).norm() / (g_ref.norm() + 1e-30))
        report.append((max(q99, l2 / 5), q99, l2, float(err.max()), name))
    report.sort(reverse=True)
    worst = [(f"q99={a:.1e}", f"l2={b:.1e}", f"max={c:.1e}", n) for _, a, b, c, n in report[:6]]
    assert report[0][0] <= tol, (k, which, n_mis, worst)
    emax = max(report, key=lambda r: r[3])
    MAX_ELEM_LOG.append((B, which, emax[3], emax[4]))
    # every single element: flat bound when the branches are teacher-forced, else one sample's share per knife edge on top
    assert emax[3] <= (MAX_ELEM_TOL if forced or n_mis == 0 else MAX_ELEM_TOL + 3.0 * n_mis / B), (k, which, n_mis, emax[3:])
    return n_mis


BRANCH_LOG = []          # (B, "vae" | "main", CE-net encoder gradients compared?) per _compare_grads call
MAX_ELEM_LOG = []        # (B, which, largest relative element error, tensor) per _compare_grads call


def _snapshot(ref):
    import copy
    return dict(model={n: w.clone() for n, w in ref.actor_critic.state_dict().items()},
                opt=copy.deepcopy(ref.optimizer.state_dict()), vae_opt=copy.deepcopy(ref.vae_optimizer.state_dict()))


def _compare_weights(k, which, pre, grads_ref, ref, alg):
    """Weights after clip + Adam, in two parts.
    (1) ENFORCED, 2e-6 abs on EVERY weight: the oracle's pre-clip gradient is written into the real gradient arena, the
        HIP optimiser (dtc_clip_adam over the optimiser's arena range, at this batch's real sizes) steps from the
        oracle's pre-step weights / Adam moments / learning rate, and the result must equal the oracle's post-step
        weights -- the clip + Adam kernel pair on real data, independent of gradient noise.  Gradient equality itself is
        _compare_grads' job.
    (2) sanity only: the weights the HIP step produced from its OWN gradients.  Adam's update lr * m / (sqrt(v) + 1e-8)
        turns fp32 rounding noise of gradients that are ~1e-8 after clipping into differences of up to 2 * lr on those
        elements (the reference drifts the same way between 1 and 8 CPU threads), so this part only bounds
        the damage: no weight off by more than 2.5 lr-steps, the error smaller than the update itself."""
    sd_ref = ref.actor_critic.state_dict()
    sd = {n: w.cpu().clone() for n, w in alg.actor_critic.state_dict().items()}
    for name, w in sd_ref.items():                      # (2)
        diff = (sd[name] - w).abs()
        upd = (w - pre["model"][name]).norm().item()
        assert float(diff.max()) <= 2.5e-3, (k, name, float(diff.max()))
        assert diff.norm().item() <= 0.75 * upd + 1e-6, (k, name, diff.norm().item(), upd)
    ac = alg.actor_critic                               # (1)
    ac.load_state_dict(pre["model"])
    alg.optimizer.load_state_dict(pre["opt"])
    alg.vae_optimizer.load_state_dict(pre["vae_opt"])
    arena = ac.arena
    arena.grad.zero_()
    for name, g in grads_ref.items():
        arena.view(arena.grad, name).copy_(g.to(DEV))
    opt = alg.vae_optimizer if which == "vae" else alg.optimizer
    opt.set_lr(5e-4 if which == "vae" else ref.learning_rate)      # the oracle adapts the LR before its optimiser step
    opt.step(alg.max_grad_norm)
    worst = max((float((ac.state_dict()[name].cpu() - w).abs().max()), name) for name, w in sd_ref.items())
    assert worst[0] <= 2e-6, (k, which, worst)


def _teacher_forced_step(k, ref, alg, idx, e1, e2):
    """Both halves of a mini-batch, each started from the oracle's exact state."""
    from oracle.ppo_ref import StepRecord
    rec = StepRecord()
    ref.capture_grads = alg.capture_grads = True
    from dtc_amd import ops
    B = idx.numel()
    forced_log = _force_oracle_signs(ref, alg)
    can_force = alg.relu_masks and all(ops.relu_mask_ok(B, w) for w in (64, 128, 512))
    _sync_from_oracle(ref, alg)
    pre = _snapshot(ref)
    ref.vae_step(idx, e1, rec)
    row, _ = alg.step_minibatch(idx, e1, e2, which="vae")
    assert not can_force or forced_log[-1] == ("vae", 7), forced_log
    _compare_scalars(k, rec, row, ref, ("recons", "vel", "kld", "height", "vae_gnorm"))
    _compare_grads(k, "vae", rec.extra["vae_grads"], ref, alg, 26, forced=can_force)
    _compare_weights(k, "vae", pre, rec.extra["vae_grads"], ref, alg)
    _sync_from_oracle(ref, alg)
    pre = _snapshot(ref)
    ref.ppo_step(idx, e2, rec)
    row, lr = alg.step_minibatch(idx, e1, e2, which="ppo")
    _compare_scalars(k, rec, row, ref, ("surrogate", "value", "entropy", "kl_mean", "gnorm"))
    assert lr == ref.learning_rate, (k, lr, ref.learning_rate)
    assert not can_force or forced_log[-1] == ("ppo", 3), forced_log
    _compare_grads(k, "main", rec.extra["grads"], ref, alg, 31, forced=can_force)
    _compare_weights(k, "main", pre, rec.extra["grads"], ref, alg)
    alg.after_forward_hook = None


def test_initialisation_matches_reference_seed(golden):
    from dtc_amd.modules import ActorCriticDecoder
    g = golden("init")
    torch.manual_seed(3)
    ac = ActorCriticDecoder(53, 1389, 12)
    sd = ac.state_dict()
    assert list(sd.keys()) == [str(k) for k in g["keys"]]
    # orthogonal_ runs a LAPACK QR whose rounding depends on the host CPU: ~1e-5 relative across machines
    sums = np.array([sd[k].double().sum().item() for k in sd])
    np.testing.assert_allclose(sums, g["sums"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(np.array([sd[k].double().abs().sum().item() for k in sd]), g["abs_sums"], rtol=1e-5)
    ac.to(DEV)
    ac.ensure_arena()
    sd2 = ac.state_dict()      # parameters now alias the flat arena: values must be unchanged
    np.testing.assert_array_equal(np.array([sd2[k].cpu().double().sum().item() for k in sd2]), sums)


def test_storage_compute_returns_and_generator():
    from oracle import gae as OG
    ref, alg = _pair(64)
    np.testing.assert_array_equal(alg.storage.returns.cpu().numpy(), ref.storage.returns.numpy())
    np.testing.assert_allclose(alg.storage.advantages.cpu().numpy(), ref.storage.advantages.numpy(), rtol=2e-6, atol=2e-6)
    perm, _, _ = S.update_noise(64, 24, 4, 1, seed=1)
    gen = alg.storage.mini_batch_generator(4, 1, indices=perm.to(DEV))
    for i, batch in enumerate(gen):
        idx = perm[i * 384:(i + 1) * 384]
        exp = ref.storage.gather(idx)
        assert len(batch) == 16 and batch[13] == (None, None) and batch[14] is None
        for j in (0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 15):
            if j == 6:
                np.testing.assert_allclose(batch[j].cpu().numpy(), exp[j].numpy(), rtol=2e-6, atol=2e-6)
            else:
                assert torch.equal(batch[j].cpu(), exp[j]), j


def test_forward_act_evaluate_vs_oracle():
    ref, alg = _pair(64)
    st = ref.storage
    f = lambda t: t.flatten(0, 1)
    g = torch.Generator().manual_seed(99)
    eps, noise = torch.randn(1536, 16, generator=g), torch.randn(1536, 12, generator=g)
    a_ref, v_ref, lp_ref, mean_ref, sig_ref = ref.act(f(st.observations), f(st.privileged_observations),
                                                      f(st.observation_histories), f(st.base_vel), eps, noise)
    ac = alg.actor_critic
    d = lambda t: f(t).to(DEV)
    actions = ac.act(d(st.observations), d(st.observation_histories), d(st.privileged_observations), None,
                     eps=eps.to(DEV), noise=noise.to(DEV))
    values = ac.evaluate(d(st.observations), d(st.privileged_observations), d(st.base_vel))
    logp = ac.get_actions_log_prob(actions)
    tol = dict(rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(ac.action_mean.cpu().numpy(), mean_ref.numpy(), **tol)
    np.testing.assert_allclose(actions.cpu().numpy(), a_ref.numpy(), **tol)
    np.testing.assert_allclose(values.cpu().numpy(), v_ref.numpy(), **tol)
    np.testing.assert_allclose(logp.cpu().numpy(), lp_ref.numpy(), rtol=1e-5, atol=1e-5)
    np.testing.assert_allclose(ac.action_std.cpu().numpy(), sig_ref.numpy(), **tol)


def test_a_diverged_env_stays_in_its_row_on_the_rollout_side():
    """ppo.py:137-155 / actor_critic_decoder.py:409-451, 504-551: rows are independent envs.  An env whose observation and privileged
    observation went inf / NaN (routine in Isaac Gym until reset_idx) must leave the other 4095 envs' actions, values and log-probs
    BIT-identical to a clean step -- PPO.act, evaluate and the deployment policy act_teacher.  (The observation history stays clean here:
    the CE-net's outlier statistics are batch-global in the reference itself, actor_critic_decoder.py:293-299.)"""
    _, alg = _pair(64)
    N = 4096
    g = torch.Generator().manual_seed(17)
    obs, priv, hist, bv = (torch.randn(N, w, generator=g).to(DEV) for w in (53, 1389, 265, 3))
    eps, noise = torch.randn(N, 16, generator=g).to(DEV), torch.randn(N, 12, generator=g).to(DEV)
    ac = alg.actor_critic

    def run(o, p):
        a = ac.act(o, hist, p, None, eps=eps, noise=noise).clone()
        v = ac.evaluate(o, p, bv).clone()
        lp = ac.get_actions_log_prob(a).clone()
        t = ac.act_teacher(o, hist, p).clone()
        return a, v, lp, t

    clean = run(obs, priv)
    ob, pb = obs.clone(), priv.clone()
    ob[7, 3], ob[7, 40] = float("nan"), float("inf")
    pb[7, 100], pb[7, 900] = float("inf"), float("nan")
    ob[2049] = float("nan")
    pb[2049] = float("nan")
    dirty = run(ob, pb)
    keep = torch.ones(N, dtype=torch.bool, device=DEV)
    keep[[7, 2049]] = False
    for c, d, name in zip(clean, dirty, ("actions", "values", "log-prob", "act_teacher")):
        assert torch.equal(c[keep], d[keep]), name
        assert bool((~torch.isfinite(d[~keep].reshape(2, -1))).any(dim=1).all()), name
    # ... and a poisoned history: the batch statistics of the outlier rule see it (as in the reference), but no other env turns non-finite
    hb = hist.clone()
    hb[11] = float("nan")
    a = ac.act(obs, hb, priv, None, eps=eps, noise=noise)
    keep[:] = True
    keep[11] = False
    assert bool(torch.isfinite(a[keep]).all()) and not bool(torch.isfinite(a[11]).all())


def test_cenet_latent_kernel_vs_torch():
    """outlier -> lower-median replacement, z, and the backward scatter, against plain torch."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(2)
    for B in (24, 384, 6000):
        mulv = torch.randn(B, 35, generator=g)
        mulv[:, 19:] = mulv[:, 19:] * 0.3 - 0.2
        eps = torch.randn(B, 16, generator=g)
        lv = mulv[:, 19:].clone().requires_grad_(True)
        mu = mulv[:, :19].clone().requires_grad_(True)
        lvf = lv.clone()
        mean, std = lvf.mean(), lvf.std()
        out = (lvf < mean - 2 * std) | (lvf > mean + 2 * std)
        med = lvf[~out].median()
        lvf[out] = med
        z = eps * torch.exp(0.5 * lvf) + mu[:, 3:]
        gz, glv = torch.randn(B, 16, generator=g), torch.randn(B, 16, generator=g)
        (z * gz).sum().backward(retain_graph=True)
        (lvf * glv).sum().backward()
        md, ed = mulv.to(DEV), eps.to(DEV)
        zd = torch.empty(B, 16, device=DEV)
        mask = torch.empty(B, 16, dtype=torch.uint8, device=DEV)
        info = torch.zeros(4, dtype=torch.int32, device=DEV)
        ws = ops.workspace(_ffi.lib().dtc_cenet_workspace(B), DEV)
        ops.cenet_latent_fwd(md, ed, zd, mask, info, ws)
        assert int(info[0]) == int(out.sum())
        assert torch.equal(mask.cpu().bool(), out)
        np.testing.assert_array_equal(md[:, 19:].cpu().numpy(), lvf.detach().numpy())
        np.testing.assert_allclose(zd.cpu().numpy(), z.detach().numpy(), rtol=1e-6, atol=1e-6)
        dmulv = torch.zeros(B, 35)
        dmulv[:, 19:] = glv
        dm = dmulv.to(DEV)
        ops.cenet_latent_bwd(dm, gz.to(DEV), ed, md, mask, info, ws)
        np.testing.assert_allclose(dm[:, 3:19].cpu().numpy(), mu.grad[:, 3:].numpy(), rtol=1e-5, atol=1e-6)
        got, exp = dm[:, 19:].cpu().numpy(), lv.grad.numpy()
        # the median's gradient lands on ONE element holding the median value (torch picks one of the
        # duplicates, if any): compare everything but that element, and the totals
        np.testing.assert_allclose(got.sum(), exp.sum(), rtol=1e-4, atol=1e-3)
        bad = np.abs(got - exp) > 1e-4 + 1e-5 * np.abs(exp)
        assert bad.sum() <= 2


def test_gradients_strict_at_filled_weights():
    """Both halves of the first mini-batch from the SAME (filled) weights -- learning rates 0 so the VAE
    step does not move them -- where all 2.3e5 + 1.1e5 ReLU decisions coincide: every parameter gradient
    of the manual backward pass agrees with torch autograd to 2e-5 of its max."""
    from oracle.ppo_ref import StepRecord
    ref, alg = _pair(64, schedule="fixed")
    ref.capture_grads = alg.capture_grads = True
    perm, e1, e2 = S.update_noise(64, 24, 4, 5, seed=123)
    idx, rec = perm[:384], StepRecord()
    _sync_from_oracle(ref, alg)
    for g in ref.vae_optimizer.param_groups + ref.optimizer.param_groups:
        g["lr"] = 0.0
    ref.learning_rate = alg.learning_rate = 0.0
    alg.vae_optimizer.set_lr(0.0)
    ref.vae_step(idx, e1[0], rec)
    alg.step_minibatch(idx, e1[0], e2[0], which="vae")
    _compare_grads(0, "vae", rec.extra["vae_grads"], ref, alg, 26, strict=True)
    ref.ppo_step(idx, e2[0], rec)
    alg.step_minibatch(idx, e1[0], e2[0], which="ppo")
    _compare_grads(0, "main", rec.extra["grads"], ref, alg, 31, strict=True)


def test_clip_adam_kernel_vs_torch_adam():
    """Same gradients in -> same update out (incl. near-zero gradients, clipping on/off, several steps)."""
    from dtc_amd import _ffi, ops
    g = torch.Generator().manual_seed(0)
    n = 100003
    p0 = torch.randn(n, generator=g)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    lr = torch.tensor([1e-3], dtype=torch.float64, device=DEV)
    gn = torch.zeros(1, device=DEV)
    ws = ops.workspace(_ffi.lib().dtc_adam_workspace(n), DEV)
    for step in range(1, 6):
        grad = torch.randn(n, generator=g) * (10.0 ** torch.randint(-9, 1, (n,), generator=g).float())
        grad *= 3.0 if step % 2 else 1e-4          # alternate clipped / unclipped
        p_ref.grad = grad.clone()
        tn = torch.nn.utils.clip_grad_norm_([p_ref], 1.0)
        opt.step()
        gd = grad.to(DEV)
        ops.clip_adam(p, gd, m, v, 1.0, lr, 0.9, 0.999, 1e-8, step, gn, ws)
        assert abs(float(gn) - float(tn)) <= 1e-5 * float(tn)
        np.testing.assert_allclose(gd.cpu().numpy(), p_ref.grad.numpy(), rtol=2e-6, atol=0)
        np.testing.assert_allclose(p.cpu().numpy(), p_ref.detach().numpy(), rtol=0, atol=2e-6)
        st = opt.state[p_ref]
        np.testing.assert_allclose(m.cpu().numpy(), st["exp_avg"].numpy(), rtol=1e-4, atol=1e-9)
        np.testing.assert_allclose(v.cpu().numpy(), st["exp_avg_sq"].numpy(), rtol=1e-4, atol=1e-14)


@pytest.mark.parametrize("kw,steps", [(dict(), 6), (dict(schedule="fixed"), 2), (dict(use_clipped_value_loss=False), 2)])
def test_update_teacher_forced_64(kw, steps):
    """BASELINE config 1 (64 envs x 24 steps): first mini-batch steps of PPO.update."""
    ref, alg = _pair(64, **kw)
    perm, e1, e2 = S.update_noise(64, 24, 4, 5, seed=123)
    mb = 384
    for k in range(steps):
        _teacher_forced_step(k, ref, alg, perm[(k % 4) * mb:(k % 4 + 1) * mb], e1[k], e2[k])


@pytest.mark.parametrize("seed,noise_seed,steps", [(4, 123, 4), (11, 321, 1)])
def test_update_teacher_forced_4096(seed, noise_seed, steps):
    """BASELINE config 2 (4096 envs x 24 steps, B = 24576): the four full-size mini-batch steps of the first epoch
    (SURVEY.md §8c G3), teacher-forced, and one step on a second rollout / noise draw.  Prints which branch of the
    CE-net gradient comparison ran (the encoder gradients are only comparable when both sides pick the same median
    element, see _compare_grads)."""
    ref, alg = _pair(4096, seed=seed)
    perm, e1, e2 = S.update_noise(4096, 24, 4, 5, seed=noise_seed)
    del BRANCH_LOG[:]
    covered = lambda: all(any(same for B, w, same in BRANCH_LOG if B == 24576 and w == which) for which in ("vae", "main"))
    k = 0
    # the CE-net encoder / latent-head gradients (128 x 128 / grouped weight-gradient kernels at B = 24576) are only
    # comparable when both sides put the median on the same element (~92 % of the calls): that branch must have RUN for
    # both optimisers -- up to three extra steps are taken until it has, then the test FAILS instead of printing
    while k < steps or (not covered() and k < steps + 3):
        _teacher_forced_step(k, ref, alg, perm[(k % 4) * 24576:(k % 4 + 1) * 24576], e1[k], e2[k])
        k += 1
    assert covered(), BRANCH_LOG
    print("largest single-element gradient errors (relative to the tensor's max):",
          sorted(((round(e, 9), w, n) for B, w, e, n in MAX_ELEM_LOG if B == 24576), reverse=True)[:4])


# Which parameter gradients a differing data-dependent decision of the forward pass can move (prefixes of the oracle's names).
# A ReLU whose sign differs sits behind layer `stack.(k-1)`: that layer's own gradient (dZ = dY * mask) and everything its
# input gradient reaches.  Inputs of the stacks: CE-net decoder <- z, mu (CE-net encoder + latent heads), l_t (terrain encoder);
# terrain decoder <- l_t; actor <- z, mu, l_t.
_UPSTREAM = {"cenet_encoder": (), "terrain_encoder": (),
             "cenet_decoder": ("vae.cenet_encoder", "vae.latent_", "vae.terrain_encoder"),
             "terrain_decoder": ("vae.terrain_encoder",)}


def _affected_by(relu_name):
    stack, k = relu_name.split(".")
    return tuple(f"vae.{stack}.{j}." for j in range(0, int(k), 2)) + _UPSTREAM[stack]


def _unforced_half(ref, alg, which, idx, eps_ref, e1, e2, rec, budget):
    """One half-step with NOTHING teacher-forced.  Asserts (a) the number of data-dependent decisions the two forwards took
    differently stays inside the fp32 knife-edge budget, (b) every gradient tensor no differing decision can reach meets the
    flat bound (q99 / L2 <= 2e-5 AND every element <= MAX_ELEM_TOL), (c) the reachable ones meet it widened by one sample's
    share (3 / B) per differing decision.  Returns (differing ReLU signs, differing outlier entries, same median element)."""
    alg.after_forward_hook = None
    if which == "vae":
        ref.vae_step(idx, eps_ref, rec)
    else:
        ref.ppo_step(idx, eps_ref, rec)
    alg.step_minibatch(idx, e1, e2, which=which)
    B = idx.numel()
    fw, tw = alg.actor_critic._fwd_ws(B), alg._train_ws(B)
    mine = {"cenet_encoder.1": fw.value("e1"), "terrain_encoder.1": fw.value("t1"), "terrain_encoder.3": fw.value("t2")}
    if which == "vae":
        mine.update({"cenet_decoder.1": tw.value("c1"), "cenet_decoder.3": tw.value("c2"), "terrain_decoder.1": tw.value("d1"), "terrain_decoder.3": tw.value("d2")})
    per_layer = {name: int(((buf.cpu() > 0) != ref.relu_masks[name]).sum()) for name, buf in mine.items()}
    decisions = sum(buf.numel() for buf in mine.values())
    n_relu = sum(per_layer.values())
    vae = ref.actor_critic.vae
    n_out = int((fw.mask.bool().cpu() != vae.last_outlier_mask).sum())
    same_median = int(fw.info[0]) == vae.last_outliers and int(fw.info[1]) == vae.last_median_index
    print(f"[un-forced {which}] B={B}: {n_relu} of {decisions} ReLU signs differ {per_layer}; {n_out} of {fw.mask.numel()} outlier "
          f"entries differ; median on the same element: {same_median}")
    assert n_relu <= budget["relu"] * decisions and n_out <= budget["outlier"] * fw.mask.numel(), (per_layer, n_out)
    touched = set()
    for name, n in per_layer.items():
        if n:
            touched.update(_affected_by(name))
    if n_out:                            # another z for those samples: everything that consumes z, and the encoder below it
        touched.update(("vae.cenet_", "vae.latent_", "vae.terrain_encoder") if which == "vae" else
                       ("vae.cenet_encoder", "vae.latent_", "vae.terrain_encoder", "actor_body", "std"))
    skip = () if same_median else ("vae.cenet_encoder", "vae.latent_var")     # the concentrated gradient sits on another element
    grads_ref = rec.extra["vae_grads" if which == "vae" else "grads"]
    arena, cap = alg.actor_critic.arena, alg.captured["vae" if which == "vae" else "main"]
    n_flat = 0
    for name, g_ref in grads_ref.items():
        if name.startswith(skip):
            continue
        g = arena.view(cap, name).cpu()
        scale = float(g_ref.abs().max()) + 1e-30
        err = ((g - g_ref).abs() / scale).reshape(-1)
        q99 = float(torch.quantile(err, 0.99)) if err.numel() > 100 else float(err.max())
        l2 = float((g - g_ref).norm() / (g_ref.norm() + 1e-30))
        reach = name.startswith(tuple(touched))
        widen = 3.0 * (n_relu + n_out) / B if reach else 0.0
        n_flat += not reach
        # a differing decision moves ONE row of its layer's weight gradient by that sample's whole contribution (a rank-one
        # term: measured 2e-3 of the tensor's max on one element), so the per-element bound holds for the unreachable tensors only
        assert max(q99, l2 / 5) <= 2e-5 + widen and (reach or float(err.max()) <= MAX_ELEM_TOL), \
            (which, name, f"q99={q99:.1e} l2={l2:.1e} max={float(err.max()):.1e}", widen, per_layer, n_out)
    return n_relu, n_out, same_median, n_flat


def test_unforced_full_size_step_stays_inside_the_knife_edge_budget():
    """BASELINE config 2, B = 24576, both half-steps of the first mini-batch with NO teacher forcing of the backward's
    branches: the two fp32 forwards may disagree on a ReLU sign only where the pre-activation is 0 within rounding (measured
    5-15 of 5.8e7 decisions per VAE step; budget 1e-6 of the decisions) and on an outlier classification only at the
    2-sigma thresholds (budget 2e-5 of the 3.9e5 entries); every gradient no differing decision reaches is held to the flat bound."""
    from oracle.ppo_ref import StepRecord
    ref, alg = _pair(4096)
    ref.capture_grads = alg.capture_grads = True
    perm, e1, e2 = S.update_noise(4096, 24, 4, 5, seed=123)
    idx, rec = perm[:24576], StepRecord()
    budget = dict(relu=1e-6, outlier=2e-5)
    _sync_from_oracle(ref, alg)
    r = _unforced_half(ref, alg, "vae", idx, e1[0], e1[0], e2[0], rec, budget)
    assert r[3] >= 2, r                 # the terrain decoder's output layer is out of every decision's reach
    _sync_from_oracle(ref, alg)
    r = _unforced_half(ref, alg, "ppo", idx, e2[0], e1[0], e2[0], rec, budget)
    assert r[3] >= 8, r                 # the critic's four layers (+ the actor's when the outlier sets coincide)


# Free-running envelope at B = 24576: step 0 is a teacher-forced step (1e-5 class); from step 1 on Adam's lr * m / (sqrt(v) + eps)
# turns the rounding noise of near-zero gradients into +-lr weight flips (the reference against ITSELF at another
# thread count grows ~10x per step), so the bound widens per step as it does in the 64-env test above.
ENVELOPE_4096 = (2e-5, 2e-3, 6e-3, 2e-2)


def test_update_free_running_4096_matches_reference_golden(golden):
    """BASELINE config 2, FREE running (weights, Adam states and the adaptive learning rate carried from step to step) against
    the four mini-batch steps captured inside the REFERENCE's own update() (tests/golden/ppo.npz u4096_*, make_golden.py):
    scalars inside the F4 envelope (the reference against itself: a 2e-8 perturbation grows ~10x per step), the four
    learning-rate decisions exactly."""
    from dtc_amd.algorithms import ppo as P
    g = golden("ppo")
    ref, alg = _pair(4096)
    perm, e1, e2 = S.update_noise(4096, 24, 4, 5, seed=123)
    cols = dict(recons=P.S_RECONS, vel=P.S_VEL, kld=P.S_KLD, height=P.S_HEIGHT, vae_gnorm=P.S_VAE_GNORM,
                surrogate=P.S_SURR, value=P.S_VALUE, entropy=P.S_ENTROPY, kl_mean=P.S_KL, gnorm=P.S_GNORM)
    worst, lrs = [], []
    for k in range(4):
        row, lr = alg.step_minibatch(perm[k * 24576:(k + 1) * 24576], e1[k], e2[k], which="both")
        lrs.append(lr)
        for key, c in cols.items():
            refv = float(g["u4096_" + key][k])
            worst.append((abs(float(row[c]) - refv) / max(1.0, abs(refv)) / ENVELOPE_4096[k], k, key, float(row[c]), refv))
    print("free-running 4096: worst error / envelope per step:", [max(w for w in worst if w[1] == k)[:3] for k in range(4)])
    assert max(worst)[0] <= 1.0, sorted(worst, reverse=True)[:6]
    assert lrs == [float(x) for x in g["u4096_lr"][:4]], (lrs, g["u4096_lr"][:4])


@pytest.mark.parametrize("n_envs,steps", [(64, 3), (4096, 1)])
def test_update_teacher_forced_with_activation_images(n_envs, steps):
    """The trainers' image chain (trainer.use_images / DTC_IMAGES=1, off by default: DESIGN.md 4.2c): hidden activations and gradients of
    the wide stacks as activation images, image-operand forward / data-gradient / weight-gradient kernels -- same teacher-forced bounds
    as the default schedule."""
    from dtc_amd import ops
    if not ops.SPLIT:
        pytest.skip("the activation-image chain is part of the split-precision path (DTC_GEMM_SPLIT=0 selected the single-pass kernels)")
    ref, alg = _pair(n_envs)
    alg.use_images = True
    perm, e1, e2 = S.update_noise(n_envs, 24, 4, 5, seed=123)
    mb = n_envs * 24 // 4
    for k in range(steps):
        _teacher_forced_step(k, ref, alg, perm[(k % 4) * mb:(k % 4 + 1) * mb], e1[k], e2[k])
    fw, tw = alg.actor_critic._fwd_ws(mb), alg._train_ws(mb)
    assert {"t1", "t2"} <= fw.live_img and "dlt" in tw.live_img, (fw.live_img, tw.live_img)       # the image chain really ran


def test_update_free_running_matches_reference_golden(golden):
    """Free-running 1 epoch x 4 mini-batches at fixed LR vs the fixture captured from the REFERENCE
    itself (u64f_*).  Not teacher-forced: Adam turns fp32 rounding noise in near-zero gradients into
    +-lr weight flips, so the trajectories separate geometrically (SURVEY.md F4 measured the reference
    against itself: 2e-8 at step 0 -> 4e-6 at step 3 -> 1e-1 at step 19); the bound widens per step."""
    from dtc_amd.algorithms import ppo as P
    g = golden("ppo")
    ref, alg = _pair(64, num_learning_epochs=1, schedule="fixed")
    perm, e1, e2 = S.update_noise(64, 24, 4, 1, seed=123)
    out, stats, lr_hist = alg.update(perm=perm.to(DEV), eps1=e1.to(DEV), eps2=e2.to(DEV), return_stats=True)
    cols = dict(recons=P.S_RECONS, vel=P.S_VEL, kld=P.S_KLD, height=P.S_HEIGHT, vae_gnorm=P.S_VAE_GNORM,
                surrogate=P.S_SURR, value=P.S_VALUE, entropy=P.S_ENTROPY, gnorm=P.S_GNORM)
    for key, c in cols.items():
        for k in range(4):
            refv = g["u64f_" + key][k]
            assert _close(float(stats[k, c]), refv, (2e-5, 3e-4, 1.5e-3, 5e-3)[k]), (key, k, float(stats[k, c]), refv)
    ret = g["u64f_update_return"]
    for i in (0, 1, 4, 5, 6):
        assert _close(out[i], ret[i], 2e-3), (i, out[i], ret[i])
    assert out[2] == 0.0 and out[3] == 0 and alg.storage.step == 0


def test_update_full_20_steps_runs_and_is_finite():
    ref, alg = _pair(64)
    out = alg.update()
    assert all(np.isfinite(float(x)) for x in out)
    assert 1e-5 <= alg.learning_rate <= 1e-2
    assert alg.last_update_stats.shape[0] == 20


def test_optimizer_state_dict_roundtrip_with_torch_adam():
    """FusedAdam <-> torch.optim.Adam state_dict layout (checkpoint compatibility)."""
    ref, alg = _pair(64)
    perm, e1, e2 = S.update_noise(64, 24, 4, 5, seed=123)
    ref.step(perm[:384], e1[0], e2[0])
    _sync_from_oracle(ref, alg)
    sd = alg.optimizer.state_dict()
    ref_sd = ref.optimizer.state_dict()
    assert set(sd["state"].keys()) == set(ref_sd["state"].keys())
    for i, st in ref_sd["state"].items():
        assert torch.allclose(sd["state"][i]["exp_avg"].cpu(), st["exp_avg"])
        assert float(sd["state"][i]["step"]) == float(st["step"])
    ref.optimizer.load_state_dict(jsonable(sd))


def jsonable(sd):
    return dict(state={k: {a: (b.cpu() if torch.is_tensor(b) else b) for a, b in v.items()} for k, v in sd["state"].items()},
                param_groups=sd["param_groups"])


def test_act_student_and_bootstrap_probability(golden):
    """`act_student` (actor_critic_decoder.py:459-502, names bound as make_golden.py:gen_student states) and
    `adapt_bootstrap_probability` (:404-407) against the reference's own outputs; tolerance 1e-5 relative / 2e-6 absolute on the
    mean action, 2e-6 on the probability (one fp32 tanh of an fp32 quotient; the sums run in double)."""
    g = golden("student")
    ref, alg = _pair(64)
    ac = alg.actor_critic
    d = S.rollout(64, 24, seed=4)
    obs, hist, priv = (d[k].flatten(0, 1)[:512] for k in ("observations", "observation_histories", "privileged_observations"))
    lidar = torch.randn(512, 512, generator=torch.Generator().manual_seed(int(g["lidar_seed"][0])))
    got = ac.act_student(obs.to(DEV), hist.to(DEV), priv.to(DEV), lidar.to(DEV))
    np.testing.assert_allclose(got.cpu().numpy(), g["mean"], rtol=1e-5, atol=2e-6)
    with torch.no_grad():
        want = ref.actor_critic.act_student(obs, hist, priv, lidar)
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-6)
    with pytest.raises(ValueError):
        ac.act_student(obs.to(DEV), hist.to(DEV), priv.to(DEV), lidar[:, :256].to(DEV))
    for i, p in enumerate(g["bootstrap_prob"]):
        r = torch.from_numpy(g[f"rew{i}"])
        assert abs(ac.adapt_bootstrap_probability(r.to(DEV)) - p) <= 2e-6, i
        assert abs(ac.adapt_bootstrap_probability(r.to(DEV).reshape(-1, 1)) - p) <= 2e-6          # (num_envs, 1) reward buffers
    assert np.isnan(ac.adapt_bootstrap_probability(torch.ones(1, device=DEV)))                  # torch.std of one value


def test_act_teacher_and_checkpoint_roundtrip(golden, tmp_path):
    """f4: `act_expert` (deployment path through memory_mlp) vs the oracle, and OnPolicyRunner.save / load: the
    dictionary has the reference's layout (tests/golden/teacher.npz) and restores model, optimiser and iteration."""
    from dtc_amd.env import ReplayEnv
    from dtc_amd.runners import OnPolicyRunner
    g = golden("teacher")
    ref, alg = _pair(64)
    d = S.rollout(64, 24, seed=4)
    obs, hist, priv = (d[k].flatten(0, 1)[:512] for k in ("observations", "observation_histories", "privileged_observations"))
    with torch.no_grad():
        want = ref.actor_critic.act_teacher(obs, hist, priv)
    got = alg.actor_critic.act_expert(dict(obs=obs.to(DEV), obs_history=hist.to(DEV), privileged_obs=priv.to(DEV)))
    np.testing.assert_allclose(got.cpu().numpy(), want.numpy(), rtol=1e-5, atol=2e-6)
    np.testing.assert_allclose(got.cpu().numpy(), g["mean"], rtol=1e-5, atol=2e-6)
    cfg = dict(runner=dict(policy_class_name="ActorCriticDecoder", algorithm_class_name="PPO", num_steps_per_env=4,
                           save_interval=10), algorithm=dict(learning_rate=1e-3), policy=dict())
    r1 = OnPolicyRunner(ReplayEnv(32, DEV), cfg, log_dir=None, device=DEV)
    r1.learn(2)
    path = str(tmp_path / "model_2.pt")
    r1.save(path, infos={"note": 1})
    ck = torch.load(path, map_location="cpu")
    assert list(ck.keys()) == [str(k) for k in g["ckpt_keys"]]
    assert list(ck["model_state_dict"].keys()) == [str(k) for k in g["model_keys"]]
    assert sorted(ck["optimizer_state_dict"].keys()) == sorted(str(k) for k in g["opt_keys"])
    assert sorted(ck["optimizer_state_dict"]["param_groups"][0].keys()) == [str(k) for k in g["group_keys"]]
    assert len(ck["optimizer_state_dict"]["param_groups"][0]["params"]) == int(g["n_group_params"][0])
    torch.optim.Adam(ref.actor_critic.parameters()).load_state_dict(ck["optimizer_state_dict"])   # torch accepts it
    r2 = OnPolicyRunner(ReplayEnv(32, DEV), cfg, log_dir=None, device=DEV)
    assert r2.load(path) == {"note": 1} and r2.current_learning_iteration == 2
    for (k, a), (_, b) in zip(r1.alg.actor_critic.state_dict().items(), r2.alg.actor_critic.state_dict().items()):
        assert torch.equal(a, b), k
    assert r2.get_inference_policy(env_t=True) == r2.alg.actor_critic.act_expert


def test_overlapped_schedule_equals_serial_schedule_bitwise():
    """Full-size update (4096 envs x 24, 20 mini-batch steps): the three-stream schedule (two compute lanes + the
    weight-gradient stream) must give bit-identical weights / statistics to the single-stream schedule -- any missing
    dependency between the lanes shows up here as a difference."""
    from dtc_amd.algorithms import PPO
    from dtc_amd.modules import ActorCriticDecoder
    d = S.rollout(4096, 24, seed=4, device=DEV)
    perm, e1, e2 = S.update_noise(4096, 24, 4, 5, seed=123)
    results = []
    for overlap in (True, False, True):
        torch.manual_seed(3)
        ac = ActorCriticDecoder(53, 1389, 12)
        alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
        alg.overlap_wgrad = alg.overlap_lanes = overlap
        alg.init_storage(4096, 24, [53], [1389], [265], [12])
        for k, v in d.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v)
        alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
        alg.storage.step = 24
        out, stats, _ = alg.update(perm.to(DEV), e1.to(DEV), e2.to(DEV), return_stats=True)
        results.append((out, stats.clone(), {k: v.clone() for k, v in ac.state_dict().items()}, alg.learning_rate))
    for other in results[1:]:
        assert results[0][0] == other[0] and results[0][3] == other[3]
        assert torch.equal(results[0][1], other[1])
        for k, v in results[0][2].items():
            assert torch.equal(v, other[2][k]), k


def test_fused_heads_loss_equals_the_five_launches():
    """dtc_ppo_heads_loss (output layers + PPO losses + their data gradients in one launch) against the unfused sequence
    (two forward GEMMs, dtc_ppo_loss, two data-gradient GEMMs) on the same policy step at full mini-batch size: every
    gradient and the four loss scalars agree to fp32 dot-product rounding (the heads' 128-term sums are accumulated in a
    different order: VALU fma chains vs MFMA)."""
    from dtc_amd.algorithms import PPO
    from dtc_amd.algorithms import ppo as P
    from dtc_amd.modules import ActorCriticDecoder
    d = S.rollout(4096, 24, seed=4, device=DEV)
    perm, e1, e2 = S.update_noise(4096, 24, 4, 5, seed=123)
    got = []
    for fuse in (True, False):
        torch.manual_seed(3)
        ac = ActorCriticDecoder(53, 1389, 12)
        alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
        alg.fuse_heads = fuse
        alg.capture_grads = True
        alg.init_storage(4096, 24, [53], [1389], [265], [12])
        for k, v in d.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v)
        alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
        B = 24576
        row, lr = alg.step_minibatch(perm[:B].to(DEV), e1[0].to(DEV), e2[0].to(DEV), which="ppo")
        fw = ac._fwd_ws(B)
        got.append((row.clone(), lr, alg.captured["main"].clone(), fw.mean.clone(), fw.val.clone()))
    (r1, lr1, g1, m1, v1), (r0, lr0, g0, m0, v0) = got
    assert lr1 == lr0
    for col in (P.S_SURR, P.S_VALUE, P.S_ENTROPY, P.S_KL, P.S_GNORM):
        assert abs(float(r1[col]) - float(r0[col])) <= 2e-6 * max(1.0, abs(float(r0[col]))), (col, float(r1[col]), float(r0[col]))
    assert float((m1 - m0).abs().max()) <= 2e-6 * max(1.0, float(m0.abs().max()))
    assert float((v1 - v0).abs().max()) <= 2e-6 * max(1.0, float(v0.abs().max()))
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 2e-5 * scale
    assert float((g1 - g0).norm()) <= 1e-5 * float(g0.norm())


def test_training_survives_model_to_calls():
    """`.to()` on the model must not orphan the optimiser's views of the parameter arena.  (a) a same-device
    `.to()` (what OnPolicyRunner.get_inference_policy(device=...) does) keeps the arena; (b) a real round trip
    cuda -> cpu -> cuda re-builds it and the optimisers are re-bound with their Adam state.  In both cases an update
    afterwards must equal the update of an untouched twin bit for bit."""
    from dtc_amd.algorithms import PPO
    from dtc_amd.modules import ActorCriticDecoder
    d = S.rollout(64, 24, seed=4, device=DEV)
    perm, e1, e2 = S.update_noise(64, 24, 4, 5, seed=123)

    def make():
        torch.manual_seed(3)
        ac = ActorCriticDecoder(53, 1389, 12)
        alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
        alg.init_storage(64, 24, [53], [1389], [265], [12])
        return ac, alg

    def run(alg):
        for k, v in d.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v)
        alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
        alg.storage.step = 24
        return alg.update(perm.to(DEV), e1.to(DEV), e2.to(DEV))

    ac0, alg0 = make()
    run(alg0), run(alg0)
    want = {k: v.clone() for k, v in ac0.state_dict().items()}
    for mode in ("same_device", "round_trip"):
        ac, alg = make()
        run(alg)
        arena = ac.arena
        if mode == "same_device":
            ac.to(DEV)
            assert ac.arena is arena
        else:
            ac.cpu()
            ac.to(DEV)
            assert ac.arena is None          # rebuilt lazily; the optimisers still point at the old buffers here
        run(alg)
        assert alg.optimizer.arena is ac.arena and alg.vae_optimizer.arena is ac.arena
        for k, v in ac.state_dict().items():
            assert torch.equal(v, want[k]), (mode, k)


def test_evaluate_before_any_act_and_chain_inputs_follow_the_caller():
    """The rollout-side layer chains are built on first use: `evaluate` alone (compute_returns on a loaded storage, as
    bench.py does) must work, and re-pointed inputs must be read from the tensors of the CURRENT call."""
    from dtc_amd.modules import ActorCriticDecoder
    torch.manual_seed(3)
    ac = ActorCriticDecoder(53, 1389, 12).to(DEV)
    d = S.rollout(128, 3, seed=2, device=DEV)
    v0 = ac.evaluate(d["observations"][0], d["privileged_observations"][0], d["base_vel"][0])
    ac.act(d["observations"][1], d["observation_histories"][1], d["privileged_observations"][1])
    v2 = ac.evaluate(d["observations"][2], d["privileged_observations"][2], d["base_vel"][2])
    v0b = ac.evaluate(d["observations"][0].clone(), d["privileged_observations"][0].clone(), d["base_vel"][0].clone())
    assert torch.equal(v0, v0b) and not torch.equal(v0, v2)


def test_graphed_rollout_step_matches_eager_kernels():
    """PPO.act replays the rollout-step kernels from a HIP graph: the value head (no random draw) must equal the eager
    launch bit for bit on fresh inputs at every replay, the sampled actions must be fresh draws with a consistent
    log-probability, and the stored transition must hold exactly what act() returned."""
    from dtc_amd.algorithms import PPO
    from dtc_amd.modules import ActorCriticDecoder
    n = 512
    torch.manual_seed(3)
    ac = ActorCriticDecoder(53, 1389, 12)
    alg = PPO(ac, learning_rate=1e-3, device=DEV)
    alg.graph_rollout = True                      # optional mode (off by default: not faster than eager launches)
    alg.init_storage(n, 4, [53], [1389], [265], [12])
    d = S.rollout(n, 4, seed=9, device=DEV)
    prev = None
    for t in range(4):
        args = (d["observations"][t], d["privileged_observations"][t], d["observation_histories"][t], d["base_vel"][t])
        actions = alg.act(*args)
        tr = alg.transition
        want_v = ac.evaluate(args[0], args[1], args[3])
        assert torch.equal(tr.values, want_v)
        mean, sigma = tr.action_mean, tr.action_sigma
        logp = torch.distributions.Normal(mean, sigma).log_prob(actions).sum(-1)
        np.testing.assert_allclose(tr.actions_log_prob.cpu().numpy(), logp.cpu().numpy(), rtol=1e-5, atol=1e-5)
        assert prev is None or not torch.equal(prev, actions)             # fresh noise at every replay
        prev = actions.clone()
        snap = {k: getattr(tr, k).clone() for k in ("actions", "values", "action_mean")}
        alg.process_env_step(d["rewards"][t, :, 0], d["dones"][t, :, 0], d["next_observations"][t], {})
        assert torch.equal(alg.storage.actions[t], snap["actions"]) and torch.equal(alg.storage.values[t], snap["values"])
        assert torch.equal(alg.storage.mu[t], snap["action_mean"])
    assert len(alg._rollout_graphs) == 1


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["decoder", "composite"])
def test_every_update_packs_its_own_rollout(kind):
    """The trainers pack the gathered rollout rows of a mini-batch into operand images ONCE per update and mini-batch slot
    (ActorCriticDecoder.packed_input, reuse=True).  The key of that reuse must change from update to update: two consecutive updates on
    DIFFERENT rollouts -- the storage is refilled in between, as a runner does -- must each read images that equal a fresh pack of
    the rollout that is in the storage at that moment (every call of packed_input is checked, all slots, both optimisation steps)."""
    from dtc_amd import h2i
    from dtc_amd.algorithms import PPO, RecurrentDecoderPPO
    from dtc_amd.modules import ActorCriticDecoder, ActorCriticDecoderRecurrent
    n = 256
    torch.manual_seed(3)
    if kind == "decoder":
        ac = ActorCriticDecoder(53, 1389, 12)
        alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
    else:
        ac = ActorCriticDecoderRecurrent(53, 1389, 12)
        alg = RecurrentDecoderPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
    alg.init_storage(n, 24, [53], [1389], [265], [12])
    g = torch.Generator(device=DEV).manual_seed(5)
    hid = [0.1 * torch.randn(24, 1, n, 512, generator=g, device=DEV) for _ in range(2)]
    orig = type(ac).packed_input
    for seed in (9, 10):
        d = S.rollout(n, 24, seed=seed, device=DEV)
        for k, v in d.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v)
        alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
        alg.storage.step = 24
        if kind == "composite":
            alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid[0]], [hid[1]]
        checks = []

        def spy(ws, name, X, idx=None, reuse=False, checks=checks):
            img = orig(ac, ws, name, X, idx, reuse)
            fresh = h2i.HImage(ws.B, X.cols, DEV).pack(X, ws.B)
            checks.append((name, bool(torch.equal(img.buf, fresh.buf))))
            return img
        ac.packed_input = spy
        try:
            alg.update()
        finally:
            del ac.packed_input
        stale = sorted({name for name, ok in checks if not ok})
        assert len(checks) >= 40 and not stale, (seed, len(checks), stale)
