"""Data-parallel PPO on the HIP path: 2 processes share cuda:0 and talk through `gloo` (two ranks cannot share a GPU
under RCCL; the collectives go through the same dtc_amd.distributed helpers either way).
Checks, after one full update on rank-specific env shards:
  * parameters, both Adam states and the learning rate are BIT-identical on the two ranks;
  * advantages were normalised with the global mean / std (== single-process normalisation of the union);
  * the result differs from a rank-local (non-DP) update, i.e. the gradient exchange really took place."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
N_PER_RANK, WORLD = 64, 2            # defaults; `n_per_rank` / `world` arguments override them per test


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _make(rank, world, full, kind="decoder", n_per_rank=N_PER_RANK, seed=3, epochs=5):
    from dtc_amd import distributed as dp
    from dtc_amd.algorithms import PPO, RecurrentDecoderPPO
    from dtc_amd.modules import ActorCriticDecoder, ActorCriticDecoderRecurrent
    dev = "cuda:0"
    N_PER_RANK = n_per_rank
    torch.manual_seed(seed)
    if kind == "composite":                       # BASELINE config 5: the 8-GPU data-parallel model
        ac = ActorCriticDecoderRecurrent(53, 1389, 12)
        alg = RecurrentDecoderPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=dev, num_learning_epochs=epochs)
    else:
        ac = ActorCriticDecoder(53, 1389, 12)
        alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=dev, num_learning_epochs=epochs)
    alg.init_storage(N_PER_RANK, 24, [53], [1389], [265], [12])
    lo, hi = dp.shard_range(N_PER_RANK * world, rank, world)
    for k, v in full.items():
        if k != "last_values":
            getattr(alg.storage, k).copy_(v[:, lo:hi].to(dev))
    alg.storage.compute_returns(full["last_values"][lo:hi].to(dev), 0.99, 0.95)
    alg.storage.step = 24
    if kind == "composite":
        g = torch.Generator().manual_seed(55)
        hid = [0.1 * torch.randn(24, 1, N_PER_RANK * world, 512, generator=g)[:, :, lo:hi].contiguous().to(dev) for _ in range(2)]
        alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid[0]], [hid[1]]
    return alg


def _worker(rank, world, port, out, kind="decoder", overlap_exchange=True, n_per_rank=N_PER_RANK, rank_seeds=False, epochs=5):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dtc_amd import distributed as dp
        from dtc_amd import synthetic as S
        torch.cuda.set_device(0)
        N_PER_RANK = n_per_rank
        dp.trace_collectives(True)
        full = S.rollout(N_PER_RANK * world, 24, seed=4)
        # rank_seeds: every rank initialises its model from ANOTHER seed -- the broadcast at optimiser construction
        # must make them start from rank 0's weights
        alg = _make(rank, world, full, kind, n_per_rank, seed=3 + (17 * rank if rank_seeds else 0), epochs=epochs)
        alg.overlap_exchange = overlap_exchange
        adv = alg.storage.advantages.cpu().clone()
        g = torch.Generator().manual_seed(100 + rank)                 # rank-local permutation and noise (§8e)
        B = N_PER_RANK * 24 // 4
        perm = torch.randperm(4 * B, generator=g)
        e1, e2 = torch.randn(4 * epochs, B, 16, generator=g), torch.randn(4 * epochs, B, 16, generator=g)
        if kind == "composite":
            alg.update(e1.cuda(), e2.cuda())          # recurrent mini-batches: env slices, no permutation
        else:
            alg.update(perm.cuda(), e1.cuda(), e2.cuda())
        log = dp.assert_same_collective_sequence()     # same ops, sizes, dtypes, stream kinds, in the same order
        out[rank] = dict(flat=alg.actor_critic.arena.flat.cpu().clone(), lr=alg.learning_rate,
                         m=alg.optimizer.exp_avg.cpu().clone(), v=alg.vae_optimizer.exp_avg_sq.cpu().clone(), adv=adv,
                         perm=perm if n_per_rank <= 64 else None, e1=e1 if n_per_rank <= 64 else None,
                         e2=e2 if n_per_rank <= 64 else None, log=log, bytes=dp.bytes_reduced(log))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["decoder", "composite"])
def test_two_ranks_stay_bit_identical_and_exchange_gradients(kind):
    from dtc_amd import synthetic as S
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, WORLD, port, out, kind)) for r in range(WORLD)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    a, b = out[0], out[1]
    assert torch.equal(a["flat"], b["flat"]) and a["lr"] == b["lr"]
    assert torch.equal(a["m"], b["m"]) and torch.equal(a["v"], b["v"])
    assert torch.isfinite(a["flat"]).all()
    # global advantage normalisation: union of the shards has mean 0 / unbiased std 1
    adv = torch.cat([a["adv"], b["adv"]], dim=1).double()
    assert abs(float(adv.mean())) < 1e-6 and abs(float(adv.std()) - 1.0) < 1e-5
    # a rank-local update on the same shard / permutation / noise ends elsewhere: the exchange happened
    full = S.rollout(N_PER_RANK * WORLD, 24, seed=4)
    solo = _make(0, 1, {k: (v[:, :N_PER_RANK] if k != "last_values" else v[:N_PER_RANK]) for k, v in full.items()}, kind)
    if kind == "composite":
        solo.update(a["e1"].cuda(), a["e2"].cuda())
    else:
        solo.update(a["perm"].cuda(), a["e1"].cuda(), a["e2"].cuda())
    assert not torch.equal(solo.actor_critic.arena.flat.cpu(), a["flat"])


@pytest.mark.parametrize("unroll", ["1", "0"])
def test_bucketed_exchange_on_the_side_stream_equals_one_exchange_after_the_join(unroll, monkeypatch):
    """PPO exchanges each gradient bucket on the weight-gradient stream as soon as its last weight gradient is queued (overlapping the rest
    of the backward pass); the result must be bit-identical to one all-reduce per optimiser step after the join -- and to ITSELF: two
    rounds of both forms (every run a fresh pair of processes), with the unrolled loss kernels (default) and with the run-time-A ones, all
    four results equal (round 5 saw 2 of 5 such comparisons differ; round 6's findings and the rule that came out of them -- ranks that
    share a device run their lanes at default priority -- are in DESIGN.md §5, the probes in tools/flake_probe.py)."""
    monkeypatch.setenv("DTC_HEADS_UNROLL", unroll)
    ctx = mp.get_context("spawn")
    results = []
    for rnd in range(2):
        for overlap in (True, False):
            out = ctx.Manager().dict()
            port = _free_port()
            procs = [ctx.Process(target=_worker, args=(r, WORLD, port, out, "decoder", overlap, N_PER_RANK, False, 3)) for r in range(WORLD)]
            for p in procs:
                p.start()
            for p in procs:
                p.join(300)
                assert p.exitcode == 0
            assert torch.equal(out[0]["flat"], out[1]["flat"])
            results.append(out[0])
    for i, other in enumerate(results[1:], 1):
        same = (torch.equal(results[0]["flat"], other["flat"]) and results[0]["lr"] == other["lr"] and
                torch.equal(results[0]["m"], other["m"]) and torch.equal(results[0]["v"], other["v"]))
        assert same, f"run {i} ({'bucketed' if i % 2 == 0 else 'joined'} exchange, round {i // 2}) differs from run 0"


def _run(world, kind="decoder", overlap=True, n_per_rank=N_PER_RANK, rank_seeds=False, timeout=600, epochs=5):
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, kind, overlap, n_per_rank, rank_seeds, epochs)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout)
        assert p.exitcode == 0
    return dict(out)


def test_four_ranks_with_different_seeds_converge_on_rank0_weights():
    """K = 4 ranks on one device; every rank seeds its model differently.  The rank-0 broadcast at optimiser
    construction + the averaged gradients keep all four bit-identical, and every rank issued the same collective
    sequence (asserted inside the workers)."""
    out = _run(4, rank_seeds=True, epochs=2)
    for r in range(1, 4):
        assert torch.equal(out[0]["flat"], out[r]["flat"]) and out[0]["lr"] == out[r]["lr"]
        assert torch.equal(out[0]["m"], out[r]["m"]) and torch.equal(out[0]["v"], out[r]["v"])
    assert torch.isfinite(out[0]["flat"]).all()
    ops = [e[0] for e in out[0]["log"]]
    assert ops[0] == "broadcast" and ops.count("all_reduce_sum") == 2              # weights once, advantage statistics once
    # per optimiser step two gradient buckets (the KL mean rides in the header of the policy step's first bucket):
    # 8 x (2 + 2) all-reduce-means and no scalar collective between the loss and Adam
    assert ops.count("all_reduce_mean") == 8 * 4
    assert all(e[1] > 1 for e in out[0]["log"] if e[0] == "all_reduce_mean")
    streams = {e[3] for e in out[0]["log"] if e[0] == "all_reduce_mean" and e[1] > 1}
    assert streams == {"side"}                                                     # buckets travel on the weight-gradient stream


def test_two_ranks_at_full_size_exchange_real_buckets():
    """K = 2 at BASELINE's 4096 envs per rank (mini-batches of 24576): the bucketed exchange at its real sizes
    (7.42 MB / 7.76 MB of gradients per VAE / policy optimiser step), ranks bit-identical afterwards."""
    out = _run(2, n_per_rank=4096, timeout=900)
    assert torch.equal(out[0]["flat"], out[1]["flat"]) and out[0]["lr"] == out[1]["lr"]
    assert torch.equal(out[0]["m"], out[1]["m"]) and torch.equal(out[0]["v"], out[1]["v"])
    assert torch.isfinite(out[0]["flat"]).all()
    # 20 mini-batches x (VAE step: encoders + decoders 1 855 245 floats, policy step: 4-float header (KL) + actor + critic +
    # std + encoders 1 940 412 floats) + 2 advantage statistics
    assert out[0]["bytes"] == 20 * 4 * (1855245 + 1940412 + 4) + 2 * 8, out[0]["bytes"]


def _rccl_worker(port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        side = torch.cuda.Stream()
        g = torch.arange(1 << 20, dtype=torch.float32, device="cuda:0")
        ev = torch.cuda.Event()
        with torch.cuda.stream(side):                     # the pattern of PPO._exchange_bucket on the real backend
            g.mul_(2.0)
            dist.all_reduce(g[1000:500000])
            g[1000:500000].mul_(1.0)
            ev.record(side)
        torch.cuda.current_stream().wait_event(ev)
        s = torch.zeros(1, dtype=torch.float64, device="cuda:0")
        dist.all_reduce(s)                                # the float64 statistics all-reduce of the storage
        dist.barrier()
        out["ok"] = bool(torch.equal(g.cpu(), torch.arange(1 << 20, dtype=torch.float32) * 2.0))
    finally:
        dist.destroy_process_group()


def test_rccl_backend_accepts_the_collective_pattern_of_the_trainer():
    """RCCL itself (backend "nccl"), one rank: an all-reduce of a slice of a flat buffer issued inside a side-stream
    context, followed by an in-place scale, an event join, a float64 all-reduce and a barrier -- the exact call pattern
    of the data-parallel trainer -- initialises and runs on this box."""
    ctx = mp.get_context("spawn")
    out = ctx.Manager().dict()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), out))
    p.start()
    p.join(300)
    assert p.exitcode == 0 and out.get("ok") is True
