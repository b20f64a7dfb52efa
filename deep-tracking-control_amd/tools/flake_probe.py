"""Run-to-run determinism probe of the decoder trainer at the data-parallel tests' size (64 envs per rank: mini-batches of 384 rows, every
kernel a few microseconds, the host far behind or ahead of the device depending on the moment).

Every optimiser step's gradient arena (after the exchange, before the clip) is kept on the device and reduced to one checksum per
parameter tensor at the end; two runs are compared step by step and the FIRST differing (step, optimiser, parameters) is printed.

    python tools/flake_probe.py solo    REPEATS    # world 1: overlapped schedule, REPEATS runs against run 0, + one serial-schedule run
    python tools/flake_probe.py dp      REPEATS    # world 2 over gloo on one device: bucketed exchange vs exchange after the join
Environment: DTC_HEADS_UNROLL etc. select the kernels (read once per process: every run is a fresh process)."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
N_PER_RANK = int(os.environ.get("PROBE_ENVS", "64"))
EPOCHS = int(os.environ.get("PROBE_EPOCHS", "3"))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _checksums(arena, snaps):
    """[(tag, {param name: int checksum})] of the captured gradient arenas."""
    out = []
    for tag, g in snaps:
        bits = g.view(torch.int32).to(torch.int64)
        row = {}
        for name, (off, n, _) in arena.offsets.items():
            row[name] = int(bits[off:off + n].sum().item())
        out.append((tag, row))
    return out


def _worker(rank, world, port, out, overlap_exchange, overlap_lanes, key, opts=None):
    opts = dict(opts or {})
    if opts.get("nodist"):                          # two processes share the device but never talk: each is DP rank r's data, world 1
        opts["as_rank"] = rank
        world = 1
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dtc_amd import distributed as dp
        from dtc_amd import synthetic as S
        from dtc_amd.algorithms import PPO
        from dtc_amd.modules import ActorCriticDecoder
        torch.cuda.set_device(0)
        dev = "cuda:0"
        from dtc_amd import _ffi, ops
        if opts.get("empty"):                        # every torch.empty / empty_like of the trainer comes back holding this value
            fillv = dict(nan=float("nan"), big=1.0e30, zero=0.0, neg=-3.0e4)[opts["empty"]]
            real_empty, real_like = torch.empty, torch.empty_like

            def _fill(t):
                if t.is_cuda and t.numel():
                    if t.is_floating_point():
                        t.fill_(fillv)
                    else:
                        t.view(torch.uint8).fill_(0 if opts["empty"] == "zero" else 0x7f)
                return t
            torch.empty = lambda *a, **k: _fill(real_empty(*a, **k))
            torch.empty_like = lambda *a, **k: _fill(real_like(*a, **k))
        sink = torch.zeros(4, device=dev)
        if opts.get("lds") is not None:              # LDS + VGPRs of every CU hold this pattern in front of EVERY launch of the library
            lib_ = _ffi.lib()
            pat = int(opts["lds"])

            def poison():
                _ffi.check(lib_.dtc_probe_poison(pat, 1024, _ffi.ptr(sink), _ffi.stream()), "dtc_probe_poison")
            for name in ("ppo_heads_loss", "vae_loss_fused", "cenet_latent_fwd", "cenet_latent_bwd"):
                def wrapk(fn):
                    def f(*a, **k):
                        poison()
                        return fn(*a, **k)
                    return f
                setattr(ops, name, wrapk(getattr(ops, name)))
            from dtc_amd import h2i as _h2i
            for name in ("linear_fwd", "linear_dgrad", "linear_fwd_chain", "linear_dgrad_chain", "wgrad_group", "linear_fwd_mse"):
                setattr(_h2i, name, wrapk(getattr(_h2i, name)))
        as_rank, as_world = opts.get("as_rank", rank), (2 if "as_rank" in opts else world)     # one process on the data of DP rank r of 2
        full = S.rollout(N_PER_RANK * as_world, 24, seed=4)
        torch.manual_seed(3)
        ac = ActorCriticDecoder(53, 1389, 12)
        alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=dev, num_learning_epochs=EPOCHS)
        alg.init_storage(N_PER_RANK, 24, [53], [1389], [265], [12])
        lo, hi = dp.shard_range(N_PER_RANK * as_world, as_rank, as_world)
        for k, v in full.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v[:, lo:hi].to(dev))
        alg.storage.compute_returns(full["last_values"][lo:hi].to(dev), 0.99, 0.95)
        alg.storage.step = 24
        alg.overlap_exchange = overlap_exchange
        if not overlap_lanes:
            alg.overlap_wgrad = alg.overlap_lanes = False
        arena = ac.ensure_arena() if hasattr(ac, "ensure_arena") else ac.arena
        if os.environ.get("PROBE_KLCOPY"):           # the trainer's form until round 6: the KL reaches the gradient header by a torch op behind the loss
            from dtc_amd.algorithms import ppo as _P
            how, cfg0 = os.environ["PROBE_KLCOPY"], alg._loss_cfg

            def cfg_without_mirror():
                c = cfg0()
                c.kl_mirror = None
                return c

            def kl_to_header(stats):
                if dp.world_size() > 1:
                    src = stats[_P.S_KL:_P.S_KL + 1]
                    if how == "memcpy":
                        arena.kl_slot.copy_(src)                        # hipMemcpyAsync, device to device, 4 bytes
                    else:
                        torch.add(src, 0.0, out=arena.kl_slot)          # the same move by an elementwise kernel
            alg._loss_cfg, alg._kl_to_header = cfg_without_mirror, kl_to_header
        snaps = []
        for tag, opt in (("vae", alg.vae_optimizer), ("main", alg.optimizer)):
            def wrap(step, tag=tag):
                def f(*a, **kw):
                    snaps.append((f"{len(snaps) // 2}:{tag}", arena.grad.clone()))
                    return step(*a, **kw)
                return f
            opt.step = wrap(opt.step)
        twice = []
        if os.environ.get("PROBE_SYNC_HEADS"):       # "before" / "after" / "both": a device-wide synchronise around the fused heads + loss launch
            from dtc_amd import ops as _ops
            real_h, where = _ops.ppo_heads_loss, os.environ["PROBE_SYNC_HEADS"]

            def synced(*a, **k):
                if where in ("before", "both"):
                    torch.cuda.synchronize()
                real_h(*a, **k)
                if where in ("after", "both"):
                    torch.cuda.synchronize()
            _ops.ppo_heads_loss = synced
        deep = []                                    # PROBE_DEEP=1: every tensor argument of the fused heads + loss launch, before and after
        if os.environ.get("PROBE_DEEP") == "1":
            real = ops.ppo_heads_loss
            names = ("Ha Hc Wa ba Wc bc act_prev std actions old_logp old_mu old_sigma advantages returns old_values idx cfg mean value "
                     "dmean dvalue dHa dHc dstd losses lr ws").split()

            def spy(*a, imgs=None):
                k = len(deep) // 2
                idx = a[15]
                gathered = {"actions", "old_logp", "old_mu", "old_sigma", "advantages", "returns", "old_values"}
                snap = lambda: {n: (t[idx] if n in gathered else t).detach().clone() for n, t in zip(names, a) if isinstance(t, torch.Tensor)}
                deep.append((f"{k}:before", snap()))
                real(*a, imgs=imgs)
                after = snap()
                twice.append((after["mean"], after["mean"], after["Ha"], after["dmean"], after["losses"], after["value"]))
                if imgs is not None:
                    for n, im in zip(("img_dHa", "img_dHc", "img_dmean", "img_dval"), imgs):
                        if im is not None:
                            after[n] = im.buf.clone()
                deep.append((f"{k}:after", after))
            ops.ppo_heads_loss = spy
        g = torch.Generator().manual_seed(100 + as_rank)
        B = N_PER_RANK * 24 // 4
        perm = torch.randperm(4 * B, generator=g)
        e1, e2 = torch.randn(4 * EPOCHS, B, 16, generator=g), torch.randn(4 * EPOCHS, B, 16, generator=g)
        _, stats, lr_hist = alg.update(perm.cuda(), e1.cuda(), e2.cuda(), return_stats=True)
        torch.cuda.synchronize()
        cs = lambda t: int(t.contiguous().reshape(-1).view(torch.uint8).to(torch.int64).mul_(torch.arange(t.numel() * t.element_size(), device=t.device) % 251 + 1).sum().item())
        deep_sums = [(tag, {n: cs(t) for n, t in d.items()}) for tag, d in deep]
        out[(key, rank)] = dict(twice=[tuple(t.cpu() for t in tt) for tt in twice], deep=deep_sums, sums=_checksums(arena, snaps), flat=int(arena.flat.view(torch.int32).to(torch.int64).sum().item()),
                                stats=stats.double().sum(dim=1).tolist(), lr=lr_hist.tolist())
    finally:
        if world > 1:
            dist.destroy_process_group()


def run(world, overlap_exchange, overlap_lanes, key, out, opts=None):
    ctx = mp.get_context("spawn")
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out, overlap_exchange, overlap_lanes, key, opts)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(600)
        assert p.exitcode == 0, (key, p.exitcode)


def inproc(reps):
    """ONE process, one rank: `reps` trainers built and run one after the other on the same inputs -- does the FIRST one of a process (fresh
    memory, lazy module loading) end anywhere else than the later ones (recycled memory)?"""
    from dtc_amd import synthetic as S
    from dtc_amd.algorithms import PPO
    from dtc_amd.modules import ActorCriticDecoder
    torch.cuda.set_device(0)
    dev = "cuda:0"
    full = S.rollout(N_PER_RANK, 24, seed=4)
    g = torch.Generator().manual_seed(100)
    B = N_PER_RANK * 24 // 4
    perm = torch.randperm(4 * B, generator=g).cuda()
    e1, e2 = torch.randn(4 * EPOCHS, B, 16, generator=g).cuda(), torch.randn(4 * EPOCHS, B, 16, generator=g).cuda()
    sums = []
    for i in range(reps):
        torch.manual_seed(3)
        ac = ActorCriticDecoder(53, 1389, 12)
        alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=dev, num_learning_epochs=EPOCHS)
        alg.init_storage(N_PER_RANK, 24, [53], [1389], [265], [12])
        for k, v in full.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v.to(dev))
        alg.storage.compute_returns(full["last_values"].to(dev), 0.99, 0.95)
        alg.storage.step = 24
        _, stats, lr = alg.update(perm, e1, e2, return_stats=True)
        sums.append((int(ac.arena.flat.view(torch.int32).to(torch.int64).sum().item()), float(stats.double().sum())))
        del alg, ac
    print("in-process repeats:", "ALL EQUAL" if len(set(sums)) == 1 else f"{len(set(sums))} distinct results", sums[:4])


def first_deep_difference(a, b):
    for (ta, ra), (tb, rb) in zip(a.get("deep", []), b.get("deep", [])):
        bad = [k for k in ra if ra[k] != rb.get(k)]
        if bad:
            return ta, bad
    return None


def first_difference(a, b):
    for (ta, ra), (tb, rb) in zip(a["sums"], b["sums"]):
        assert ta == tb
        bad = [k for k in ra if ra[k] != rb[k]]
        if bad:
            return ta, bad
    if a["flat"] != b["flat"]:
        return "weights", []
    return None


def main():
    mode, reps = sys.argv[1], int(sys.argv[2])
    if mode == "inproc":
        return inproc(reps)
    out = mp.get_context("spawn").Manager().dict()
    world = 2 if mode == "dp" else 1
    plan = []
    if mode == "dp":
        for i in range(reps):
            plan += [(f"overlap{i}", True, True), (f"joined{i}", False, True)]
    elif mode == "poison":
        # one rank, the overlapped schedule; what the kernels find in memory they never wrote differs from run to run ON PURPOSE
        plan = [("ref", True, True, {}), ("empty_nan", True, True, dict(empty="nan")), ("empty_big", True, True, dict(empty="big")),
                ("empty_neg", True, True, dict(empty="neg")), ("lds_nan", True, True, dict(lds=0x7fc00000)),
                ("lds_big", True, True, dict(lds=0x7e967699)), ("lds_ones", True, True, dict(lds=0x3f800000)),
                ("lds_int", True, True, dict(lds=0x00010001))][:max(2, reps)]
    elif mode == "pair":
        world = 2
        plan = [(f"pair{i}", True, True, dict(nodist=True)) for i in range(reps)]
    elif mode == "asrank":
        r = int(os.environ.get("PROBE_AS_RANK", "1"))
        plan = [(f"rank{r}data_{i}", True, True, dict(as_rank=r)) for i in range(reps)]
        plan += [(f"rank{r}data_lds_nan", True, True, dict(as_rank=r, lds=0x7fc00000)), (f"rank{r}data_empty_nan", True, True, dict(as_rank=r, empty="nan")),
                 (f"rank{r}data_serial", True, False, dict(as_rank=r))]
    else:
        plan = [(f"lanes{i}", True, True) for i in range(reps)] + [("serial0", True, False), ("serial1", True, False)]
    plan = [p if len(p) == 4 else p + (None,) for p in plan]
    for key, ox, ol, opts in plan:
        run(world, ox, ol, key, out, opts)
    plan = [p[:3] for p in plan]
    ref_key = plan[0][0]
    n_bad = 0
    for key, _, _ in plan:
        for r in range(world):
            if world > 1 and out[(key, 0)]["flat"] != out[(key, r)]["flat"]:
                print(f"{key}: RANKS DIFFER (rank {r})")
            d = first_difference(out[(ref_key, r)], out[(key, r)])
            stats_same = out[(ref_key, r)]["stats"] == out[(key, r)]["stats"]
            lr_same = out[(ref_key, r)]["lr"] == out[(key, r)]["lr"]
            if d is None and stats_same and lr_same:
                print(f"{key} rank {r}: identical to {ref_key}")
            else:
                n_bad += 1
                first_stat = next((i for i, (x, y) in enumerate(zip(out[(ref_key, r)]["stats"], out[(key, r)]["stats"])) if x != y), None)
                first_lr = next((i for i, (x, y) in enumerate(zip(out[(ref_key, r)]["lr"], out[(key, r)]["lr"])) if x != y), None)
                dd = first_deep_difference(out[(ref_key, r)], out[(key, r)])
                if dd is not None:
                    print(f"   heads + loss launch: first difference at {dd[0]} in {dd[1]}")
                    k = int(dd[0].split(":")[0])
                    ta, tb = out[(ref_key, r)]["twice"], out[(key, r)]["twice"]
                    if k < len(ta) and k < len(tb):
                        (m1a, m2a, haa, dma, la, va), (m1b, m2b, hab, dmb, lb, vb) = ta[k], tb[k]
                        rows = (m1a != m1b).any(dim=1).nonzero().flatten().tolist()
                        print(f"   step {k}: mean rows that differ between the runs: {len(rows)} of {m1a.shape[0]}: {rows[:24]}; "
                              f"max |diff| {float((m1a - m1b).abs().max()):.3e} (max |mean| {float(m1a.abs().max()):.3e}); "
                              f"second launch in the same step == first: ref run {bool(torch.equal(m1a, m2a))}, this run {bool(torch.equal(m1b, m2b))}; "
                              f"second launches equal across runs: {bool(torch.equal(m2a, m2b))}; Ha equal: {bool(torch.equal(haa, hab))}")
                        if rows:
                            cols = (m1a[rows[0]] != m1b[rows[0]]).nonzero().flatten().tolist()
                            print(f"   row {rows[0]}: columns {cols}; ref {m1a[rows[0]].tolist()}; this {m1b[rows[0]].tolist()}")
                        print(f"   losses ref {la.tolist()} this {lb.tolist()}; dmean rows differing {int((dma != dmb).any(dim=1).sum())}; value equal {bool(torch.equal(va, vb))}")
                print(f"{key} rank {r}: DIFFERS from {ref_key}: first gradient difference at step {d[0] if d else None} "
                      f"in {d[1][:8] if d else None} ({len(d[1]) if d else 0} tensors); first statistics row that differs: {first_stat}; "
                      f"first lr that differs: {first_lr}")
    print(f"SUMMARY mode={mode} reps={reps} unroll={os.environ.get('DTC_HEADS_UNROLL', '0')} "
          f"serialize={os.environ.get('AMD_SERIALIZE_KERNEL', '0')}: {n_bad} differing (run, rank) pairs of {len(plan) * world}")


if __name__ == "__main__":
    main()
