"""Data-parallel path on CPU: 2 processes, `gloo` backend.

The collectives of the hot path live in dtc_amd/distributed.py (device agnostic).  Here every rank
holds an env shard, computes its local quantities with the CPU oracle (the HIP kernels need a GPU),
pushes them through the SAME helpers PPO / RolloutStorage call on the GPU, and the result is compared
with the single-process computation on the union of the shards:
  * gradient bucket: mean over ranks of the shard gradients == K-shard emulation (G7);
  * advantage normalisation: two scalar all-reduces reproduce the global mean / unbiased std;
  * KL mean: every rank ends with the same learning rate;
  * parameters stay bit-identical across ranks after the update.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dtc_amd import distributed as dp
from dtc_amd import synthetic as S

N_PER_RANK = 16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _flat_grads(ac):
    return torch.cat([p.grad.reshape(-1) if p.grad is not None else torch.zeros(p.numel()) for p in ac.parameters()])


def _shard_step(rank, world, results):
    """One VAE + PPO mini-batch on this rank's shard with DP semantics; returns what the test compares."""
    from oracle import gae as OG
    from oracle import ppo_ref as OP
    torch.set_num_threads(1)
    torch.manual_seed(3)
    ac = OP.fill_parameters_(OP.RefActorCriticDecoder(), 11)
    alg = OP.RefPPO(ac, learning_rate=1e-3, entropy_coef=0.003)
    N = N_PER_RANK
    full = S.rollout(N * world, 24, seed=4)
    lo, hi = dp.shard_range(N * world, rank, world)
    alg.init_storage(N, 24)
    for k, v in full.items():
        if k != "last_values":
            getattr(alg.storage, k).copy_(v[:, lo:hi])
    # --- advantage normalisation through the two scalar all-reduces
    sq = lambda k: getattr(alg.storage, k).squeeze(-1).numpy()
    ret = OG.gae_scan(sq("rewards"), sq("values"), sq("dones"), full["last_values"][lo:hi, 0].numpy())
    adv = torch.from_numpy(ret - sq("values")).double()
    stats = torch.zeros(2, dtype=torch.float64)
    stats[0] = adv.sum()
    dp.allreduce_sum_(stats[0:1])
    count = float(adv.numel() * world)
    mean = stats[0] / count
    stats[1] = ((adv - mean) ** 2).sum()
    dp.allreduce_sum_(stats[1:2])
    std = torch.sqrt(stats[1] / (count - 1))
    adv_n = ((adv - mean) / (std + 1e-8)).float()
    alg.storage.returns.copy_(torch.from_numpy(ret).unsqueeze(-1))
    alg.storage.advantages.copy_(adv_n.unsqueeze(-1))
    # --- one mini-batch: local backward, gradient bucket all-reduce (mean), identical optimiser step
    perm, e1, e2 = S.update_noise(N, 24, 4, 5, seed=123 + rank)       # rank-local permutation / noise
    idx = perm[:N * 24 // 4]
    alg.capture_grads = True
    rec = OP.StepRecord()
    # VAE half with averaged gradients
    alg2 = alg
    _dp_half(alg2, "vae", idx, e1[0], rec)
    _dp_half(alg2, "ppo", idx, e2[0], rec)
    flat = torch.cat([p.detach().reshape(-1) for p in ac.parameters()])
    results[rank] = dict(adv=adv_n.numpy(), mean=float(mean), std=float(std), lr=alg.learning_rate,
                         params=flat.numpy(), kl=rec.kl_mean, vae_grad=rec.extra["dp_vae_grad"].numpy(),
                         main_grad=rec.extra["dp_main_grad"].numpy())


def _dp_half(alg, which, idx, eps, rec):
    """The oracle's half-step with the data-parallel hooks inserted where PPO._allreduce_grads /
    dtc_lr_adapt sit on the GPU path: local backward -> bucket all-reduce(mean) -> clip -> Adam."""
    import torch.nn as nn
    ac = alg.actor_critic
    clip = nn.utils.clip_grad_norm_

    def averaged_clip(params, max_norm, *a, **k):
        params = [p for p in params if p.grad is not None]
        bucket = torch.cat([p.grad.reshape(-1) for p in params])
        dp.allreduce_mean_(bucket)
        rec.extra["dp_vae_grad" if which == "vae" else "dp_main_grad"] = bucket.clone()
        off = 0
        for p in params:
            p.grad.copy_(bucket[off:off + p.numel()].view_as(p.grad))
            off += p.numel()
        return clip(params, max_norm, *a, **k)

    nn.utils.clip_grad_norm_ = averaged_clip
    try:
        if which == "vae":
            alg.vae_step(idx, eps, rec)
        else:
            # KL statistic averaged over ranks BEFORE the learning-rate rule (same branch everywhere)
            orig_mean = torch.mean

            def kl_mean(x, *a, **k):
                m = orig_mean(x, *a, **k)
                if torch.is_inference_mode_enabled():          # only the KL mean is taken in inference mode
                    m = m.clone()
                    dp.allreduce_mean_(m)
                return m
            torch.mean = kl_mean
            try:
                alg.ppo_step(idx, eps, rec)
            finally:
                torch.mean = orig_mean
    finally:
        nn.utils.clip_grad_norm_ = clip


def _worker(rank, world, port, ret_dict):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        assert dp.world_size() == world and dp.rank() == rank
        _shard_step(rank, world, ret_dict)
    finally:
        dist.destroy_process_group()


@pytest.fixture(scope="module", params=[2, 8])      # K = 2 and K = 8 shards
def dp_results(request):
    world = request.param
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    out = dict(ret)
    out["world"] = world
    return out


def test_single_process_helpers_are_noops():
    t = torch.ones(4)
    assert dp.world_size() == 1 and dp.rank() == 0
    assert torch.equal(dp.allreduce_mean_(t.clone()), t) and torch.equal(dp.allreduce_sum_(t.clone()), t)
    assert dp.shard_range(4096, 3, 8) == (1536, 2048)


def test_advantage_normalisation_is_global(dp_results):
    from oracle import gae as OG
    WORLD = dp_results["world"]
    full = S.rollout(N_PER_RANK * WORLD, 24, seed=4)
    sq = lambda k: full[k].squeeze(-1).numpy()
    ret, adv = OG.compute_returns(sq("rewards"), sq("values"), sq("dones"), full["last_values"][:, 0].numpy())
    got = np.concatenate([dp_results[r]["adv"] for r in range(WORLD)], axis=1)
    np.testing.assert_allclose(got, adv, rtol=2e-6, atol=2e-6)
    assert all(dp_results[r]["mean"] == dp_results[0]["mean"] and dp_results[r]["std"] == dp_results[0]["std"] for r in range(WORLD))


def test_ranks_stay_identical_after_a_step(dp_results):
    a = dp_results[0]
    for r in range(1, dp_results["world"]):
        b = dp_results[r]
        np.testing.assert_array_equal(a["params"], b["params"])           # same averaged gradient, same LR
        assert a["lr"] == b["lr"] and a["kl"] == b["kl"]
        np.testing.assert_array_equal(a["main_grad"], b["main_grad"])
        np.testing.assert_array_equal(a["vae_grad"], b["vae_grad"])


def test_gradient_bucket_equals_shard_emulation(dp_results):
    """K-shard emulation in ONE process (G7): average of the per-shard gradients."""
    from oracle import gae as OG
    from oracle import ppo_ref as OP
    torch.set_num_threads(1)
    WORLD = dp_results["world"]
    full = S.rollout(N_PER_RANK * WORLD, 24, seed=4)
    sq = lambda k: full[k].squeeze(-1).numpy()
    ret, adv = OG.compute_returns(sq("rewards"), sq("values"), sq("dones"), full["last_values"][:, 0].numpy())
    grads = []
    for r in range(WORLD):
        torch.manual_seed(3)
        ac = OP.fill_parameters_(OP.RefActorCriticDecoder(), 11)
        alg = OP.RefPPO(ac, learning_rate=0.0, entropy_coef=0.003, schedule="fixed")
        for g in alg.vae_optimizer.param_groups:
            g["lr"] = 0.0
        alg.init_storage(N_PER_RANK, 24)
        lo, hi = dp.shard_range(N_PER_RANK * WORLD, r, WORLD)
        for k, v in full.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v[:, lo:hi])
        alg.storage.returns.copy_(torch.from_numpy(ret[:, lo:hi]).unsqueeze(-1))
        alg.storage.advantages.copy_(torch.from_numpy(adv[:, lo:hi]).unsqueeze(-1))
        perm, e1, e2 = S.update_noise(N_PER_RANK, 24, 4, 5, seed=123 + r)
        rec = OP.StepRecord()
        alg.capture_grads = True
        alg.vae_step(perm[:N_PER_RANK * 24 // 4], e1[0], rec)
        grads.append(torch.cat([g.reshape(-1) for g in rec.extra["vae_grads"].values()]))
    expect = torch.stack(grads).mean(0).numpy()
    np.testing.assert_allclose(dp_results[0]["vae_grad"], expect, rtol=1e-5, atol=1e-7)
