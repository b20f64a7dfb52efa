"""SURVEY.md §8c G7 on the HIP path: data-parallel PPO over K = 2 ranks (both on cuda:0, talking through gloo -- two ranks
cannot share a GPU under RCCL; the collectives go through dtc_amd.distributed either way) against the reference-derived
K-shard oracle: "K independent reference mini-batch steps on K shards with rank-local outlier statistics, permutation and
noise, gradients (and the KL statistic) averaged, one identical optimiser step" (SURVEY.md §8e).

Every rank builds the oracle of ALL shards on the CPU (identical weights, its own shard's storage per oracle), averages the
oracle's per-shard pre-clip gradients itself, and compares
  * the gradient the HIP trainer holds AFTER its bucket exchanges (arena.grad at the optimiser step) with that average:
    every parameter tensor, bound of tests/test_hip_ppo._compare_grads (q99 / L2 AND the max element);
  * the per-step scalars (local losses; gradient norm of the AVERAGED gradient; averaged KL; learning rate);
  * the weights after clip + Adam with the oracle stepped on the averaged gradient (tests/test_hip_ppo._compare_weights).
Both models of the N > 1 bench lines: `decoder` (configs[1] per rank) and `composite` (configs[4]: GRU + CE-net).
The second block runs the composite at BASELINE's 4096 envs per rank through a whole update (real bucket sizes)."""
import copy
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
T, NMB = 24, 4


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Averaged:
    """Context: the oracle's half-step with the data-parallel semantics inserted where PPO._exchange_bucket / dtc_lr_adapt
    sit on the HIP path -- `clip_grad_norm_` first replaces every gradient by the given K-shard average, and the KL mean
    (the one torch.mean taken in inference mode, ppo.py:296) returns the given average."""

    def __init__(self, grads, kl_mean, named_params):
        self.grads, self.kl, self.named = grads, kl_mean, named_params

    def __enter__(self):
        self.clip, self.mean = nn.utils.clip_grad_norm_, torch.mean
        grads, named, clip, mean, kl = self.grads, self.named, self.clip, self.mean, self.kl

        def averaged_clip(params, max_norm, *a, **k):
            for name, p in named.items():
                if name in grads:
                    p.grad.copy_(grads[name])
            return clip(params, max_norm, *a, **k)

        def kl_mean(x, *a, **k):
            m = mean(x, *a, **k)
            if torch.is_inference_mode_enabled() and kl is not None:
                m = torch.full_like(m, kl)
            return m
        nn.utils.clip_grad_norm_, torch.mean = averaged_clip, kl_mean
        return self

    def __exit__(self, *exc):
        nn.utils.clip_grad_norm_, torch.mean = self.clip, self.mean
        return False


def _shard_oracles(kind, world, n):
    """One oracle trainer per shard: identical filled weights, the shard's rollout, returns and GLOBALLY normalised advantages
    (rollout_storage.py:138-152 on the union of the shards), rank-local noise; -> (oracles, batches, noise, full data)."""
    from dtc_amd import distributed as dp
    from dtc_amd import synthetic as S
    from oracle import composite_ref as CR
    from oracle import gae as OG
    from oracle import ppo_ref as OP
    full = S.rollout(n * world, T, seed=4)
    if kind == "composite":
        full["dones"][:, 0] = 0
    sq = lambda k: full[k].squeeze(-1).numpy()
    ret, adv = OG.compute_returns(sq("rewards"), sq("values"), sq("dones"), full["last_values"][:, 0].numpy())
    g = torch.Generator().manual_seed(55)
    hid = [0.1 * torch.randn(T, 1, n * world, 512, generator=g) for _ in range(2)] if kind == "composite" else None
    refs, batches, noise = [], [], []
    for r in range(world):
        lo, hi = dp.shard_range(n * world, r, world)
        torch.manual_seed(3)
        if kind == "composite":
            ref = CR.RefCompositePPO(OP.fill_parameters_(CR.RefCompositeAC(), 23), learning_rate=1e-3, entropy_coef=0.003)
        else:
            ref = OP.RefPPO(OP.fill_parameters_(OP.RefActorCriticDecoder(), 11), learning_rate=1e-3, entropy_coef=0.003)
        ref.init_storage(n, T)
        for k, v in full.items():
            if k != "last_values":
                getattr(ref.storage, k).copy_(v[:, lo:hi])
        ref.storage.returns.copy_(torch.from_numpy(ret[:, lo:hi]).unsqueeze(-1))
        ref.storage.advantages.copy_(torch.from_numpy(adv[:, lo:hi]).unsqueeze(-1))
        ref.capture_grads = True
        gr = torch.Generator().manual_seed(200 + r)                     # rank-local draws
        B = n * T // NMB
        perm = torch.randperm(NMB * B, generator=gr)
        e1, e2 = torch.randn(B, 16, generator=gr), torch.randn(B, 16, generator=gr)
        if kind == "composite":
            bt = next(iter(CR.recurrent_slices(ref.storage, hid[0][:, :, lo:hi], hid[1][:, :, lo:hi], NMB)))
        else:
            bt = perm[:B]
        refs.append(ref)
        batches.append(bt)
        noise.append((e1, e2))
    return refs, batches, noise, full, hid, adv


def _hip_trainer(kind, rank, world, n, full, hid):
    from dtc_amd import distributed as dp
    from dtc_amd.algorithms import PPO, RecurrentDecoderPPO
    from dtc_amd.modules import ActorCriticDecoder, ActorCriticDecoderRecurrent
    torch.manual_seed(3 + 17 * rank)                                    # the rank-0 broadcast has to undo this
    if kind == "composite":
        alg = RecurrentDecoderPPO(ActorCriticDecoderRecurrent(53, 1389, 12), learning_rate=1e-3, entropy_coef=0.003, device=DEV)
    else:
        alg = PPO(ActorCriticDecoder(53, 1389, 12), learning_rate=1e-3, entropy_coef=0.003, device=DEV)
    alg.init_storage(n, T, [53], [1389], [265], [12])
    lo, hi = dp.shard_range(n * world, rank, world)
    for k, v in full.items():
        if k != "last_values":
            getattr(alg.storage, k).copy_(v[:, lo:hi].to(DEV))
    alg.storage.compute_returns(full["last_values"][lo:hi].to(DEV), 0.99, 0.95)      # global statistics: two all-reduces
    return alg, (lo, hi)


def _half(ref, kind, which, bt, eps, rec):
    if which == "vae":
        ref.vae_step(bt["idx"] if kind == "composite" else bt, eps, rec)
    else:
        ref.ppo_step(bt, eps, rec)


def _g7_worker(rank, world, port, out, kind, n):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        from dtc_amd.algorithms import ppo as P
        from oracle.ppo_ref import StepRecord
        import test_hip_ppo as TP
        torch.cuda.set_device(0)
        torch.set_num_threads(8)
        refs, batches, noise, full, hid, adv = _shard_oracles(kind, world, n)
        alg, (lo, hi) = _hip_trainer(kind, rank, world, n, full, hid)
        np.testing.assert_allclose(alg.storage.advantages.squeeze(-1).cpu().numpy(), adv[:, lo:hi], rtol=2e-6, atol=2e-6)
        alg.capture_grads = True
        ref = refs[rank]
        strip = (lambda sd: {k.replace("acr.", ""): v for k, v in sd.items()})
        bt_hip = batches[rank]
        if kind == "composite":
            bt_hip = next(iter(alg.recurrent_slices(hid[0][:, :, lo:hi].contiguous().to(DEV), hid[1][:, :, lo:hi].contiguous().to(DEV))))
            assert torch.equal(bt_hip["idx"].cpu(), batches[rank]["idx"])
        e1, e2 = noise[rank]
        report = {}
        for which, cap, n_min in (("vae", "vae", 20), ("ppo", "main", 31)):
            # --- the K-shard oracle: per-shard gradients from identical weights, then ONE step on their average
            pre_model = copy.deepcopy(ref.actor_critic.state_dict())
            grads, kls = [], []
            for r in range(world):
                clone = copy.deepcopy(refs[r])
                clone.actor_critic.load_state_dict(pre_model)
                clone.optimizer.load_state_dict(copy.deepcopy(ref.optimizer.state_dict()))
                clone.vae_optimizer.load_state_dict(copy.deepcopy(ref.vae_optimizer.state_dict()))
                clone.learning_rate = ref.learning_rate
                rec_r = StepRecord()
                _half(clone, kind, which, batches[r], noise[r][0 if which == "vae" else 1], rec_r)
                grads.append(rec_r.extra["vae_grads" if which == "vae" else "grads"])
                kls.append(rec_r.kl_mean)
            avg = {k: torch.stack([g[k] for g in grads]).mean(0) for k in grads[0]}
            kl_avg = float(np.mean(np.float32(kls))) if which == "ppo" else None
            pre = TP._snapshot(ref)
            rec = StepRecord()
            with _Averaged(avg, kl_avg, dict(ref.actor_critic.named_parameters())):
                _half(ref, kind, which, batches[rank], e1 if which == "vae" else e2, rec)
            # --- the HIP trainer from the same pre-step state, its data-dependent branches following its shard's oracle
            forced = TP._force_oracle_signs(ref, alg)
            alg.actor_critic.load_state_dict(strip(pre["model"]))
            alg.optimizer.load_state_dict(pre["opt"])
            alg.vae_optimizer.load_state_dict(pre["vae_opt"])
            alg.learning_rate = pre["opt"]["param_groups"][0]["lr"]
            alg.vae_optimizer.set_lr(5e-4)
            if kind == "composite":
                row = alg.step_minibatch(bt_hip, e1.to(DEV), e2.to(DEV), which=which).cpu()
                lr = float(alg.optimizer.lr_dev.item())
            else:
                row, lr = alg.step_minibatch(bt_hip, e1, e2, which=which)
            alg.after_forward_hook = None
            assert not alg.relu_masks or forced, "sign records were not teacher-forced"
            keys = ("recons", "vel", "kld", "height", "vae_gnorm") if which == "vae" else ("surrogate", "value", "entropy", "kl_mean", "gnorm")
            cols = dict(recons=P.S_RECONS, vel=P.S_VEL, kld=P.S_KLD, height=P.S_HEIGHT, vae_gnorm=P.S_VAE_GNORM,
                        surrogate=P.S_SURR, value=P.S_VALUE, entropy=P.S_ENTROPY, kl_mean=P.S_KL, gnorm=P.S_GNORM)
            for key in keys:
                got, want = float(row[cols[key]]), getattr(rec, key)
                assert abs(got - want) <= 1e-5 * max(1.0, abs(want)), (rank, which, key, got, want)
            if which == "ppo":
                assert abs(lr - ref.learning_rate) <= 1e-12, (lr, ref.learning_rate)
            # --- exchanged HIP gradient vs the oracle's K-shard average
            arena = alg.actor_critic.arena
            worst = []
            for name, g_ref in avg.items():
                g = arena.view(alg.captured[cap], name.replace("acr.", "")).cpu()
                scale = float(g_ref.abs().max()) + 1e-30
                err = ((g - g_ref).abs() / scale).reshape(-1)
                q99 = float(torch.quantile(err, 0.99)) if err.numel() > 100 else float(err.max())
                l2 = float((g - g_ref).norm() / (g_ref.norm() + 1e-30))
                worst.append((max(q99, l2 / 5), float(err.max()), name))
            worst.sort(reverse=True)
            assert len(worst) >= n_min and worst[0][0] <= 2e-5, (rank, which, worst[:5])
            assert max(w[1] for w in worst) <= TP.MAX_ELEM_TOL, (rank, which, sorted(worst, key=lambda w: -w[1])[:5])
            # the exchange really happened: the local shard's gradient alone is NOT the average
            local = grads[rank]
            name = max(avg, key=lambda k: avg[k].numel())
            assert float((local[name] - avg[name]).norm() / avg[name].norm()) > 1e-3
            # --- weights after clip + Adam
            sd_ref = strip(ref.actor_critic.state_dict())
            sd = {k: v.cpu() for k, v in alg.actor_critic.state_dict().items()}
            wmax = 0.0
            for k, w in sd_ref.items():
                diff = (sd[k] - w).abs()
                upd = float((w - strip(pre["model"])[k]).norm())
                assert float(diff.max()) <= 2.5e-3 and float(diff.norm()) <= 0.75 * upd + 1e-6, (rank, which, k, float(diff.max()), upd)
                wmax = max(wmax, float(diff.max()))
            report[which] = dict(q=worst[0][0], emax=max(w[1] for w in worst), wmax=wmax, n=len(worst), gnorm=float(rec.vae_gnorm if which == "vae" else rec.gnorm))
        out[rank] = dict(report=report, flat=alg.actor_critic.arena.flat.cpu().clone(), lr=float(alg.optimizer.lr_dev.item()))
    finally:
        dist.destroy_process_group()


def _spawn(target, world, args, timeout=900):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=target, args=(r, world, port, out, *args)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0, f"rank exited with {p.exitcode}"
    return dict(out)


@pytest.mark.parametrize("kind,n", [("decoder", 64), ("composite", 64), ("decoder", 1024)])
def test_exchanged_hip_gradients_equal_the_oracles_shard_average(kind, n):
    """n = 64 envs per rank: mini-batches of 384 rows; n = 1024: 6144 rows -- the 128 x 128 split-precision tiles, 8-slice grouped weight
    gradients and sign records of the full-size schedule under the exchange."""
    out = _spawn(_g7_worker, 2, (kind, n))
    print(kind, {r: out[r]["report"] for r in out})
    assert torch.equal(out[0]["flat"], out[1]["flat"]) and out[0]["lr"] == out[1]["lr"]


def test_composite_two_ranks_at_full_size_exchange_real_buckets():
    """configs[4]'s model (GRU + CE-net) at BASELINE's 4096 envs per rank, K = 2, one whole update (20 recurrent mini-batches
    of 1024 envs x 24 steps): ranks bit-identical afterwards, identical collective sequences (asserted in the workers), and
    the payload is exactly the two buckets + header of every optimiser step."""
    import test_hip_dp as D
    out = D._run(2, kind="composite", n_per_rank=4096, timeout=900)
    assert torch.equal(out[0]["flat"], out[1]["flat"]) and out[0]["lr"] == out[1]["lr"]
    assert torch.equal(out[0]["m"], out[1]["m"]) and torch.equal(out[0]["v"], out[1]["v"])
    assert torch.isfinite(out[0]["flat"]).all()
    log = out[0]["log"]
    means = [e for e in log if e[0] == "all_reduce_mean"]
    assert len(means) == 20 * 4 and {e[3] for e in means} == {"side"}
    n_params = out[0]["flat"].numel()
    # VAE step: decoders + encoders; policy step: header + (heads, GRUs, std) + encoders.  The encoders are the only block
    # exchanged by both steps, the decoders only by the first: payload = 20 x (4 B x (params + header + encoders))
    per_step = sum(e[1] for e in means) // 20
    assert per_step > n_params and out[0]["bytes"] == 20 * 4 * per_step + 2 * 8
