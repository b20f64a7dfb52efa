"""Run-to-run stress of dtc_ppo_heads_loss_img on FIXED inputs: `reps` launches, every output compared with the first launch's on the
device (no host synchronisation in the loop).  --noise N: N child processes keep the same GPU busy with other kernels (the data-parallel
tests put two ranks on one device); --streams: a second stream of this process runs elementwise kernels beside the launches.

    DTC_HEADS_UNROLL=1 python tools/heads_stress.py 384 20000 --noise 1"""
import os
import subprocess
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, h2i, ops  # noqa: E402

DEV = "cuda:0"


def many_streams(q):
    """q streams; NOISE_PRIO = high | normal | mixed (alternating: each priority has its own hardware queues)."""
    how = os.environ.get("NOISE_PRIO", "mixed")
    return [torch.cuda.Stream(priority=-1 if (how == "high" or (how == "mixed" and i % 2 == 0)) else 0) for i in range(q)]


def noise_loop():
    x = torch.randn(2048, 2048, device=DEV)
    y = torch.randn(1 << 22, device=DEV)
    q = int(os.environ.get("NOISE_QUEUES", "0"))
    streams = many_streams(q)
    small = [torch.randn(384, 512, device=DEV) for _ in streams]
    w = torch.randn(512, 512, device=DEV) * 0.04
    heavy = os.environ.get("NOISE_HEAVY", "1") == "1"
    t0 = time.time()
    while time.time() - t0 < float(os.environ.get("NOISE_SECONDS", "600")):
        for _ in range(50):
            if heavy:
                x = torch.tanh(x @ x * 1e-3)
                y = torch.sin(y) * 1.0001
            for i, s_ in enumerate(streams):            # bursts of short kernels (a small GEMM + an elementwise op) on the queues
                with torch.cuda.stream(s_):
                    small[i] = torch.tanh(small[i] @ w)
        torch.cuda.synchronize()
        if not heavy:
            time.sleep(0.0002)


def main():
    if sys.argv[1] == "noise":
        return noise_loop()
    B, reps = int(sys.argv[1]), int(sys.argv[2])
    n_noise = int(sys.argv[sys.argv.index("--noise") + 1]) if "--noise" in sys.argv else 0
    nq = int(sys.argv[sys.argv.index("--queues") + 1]) if "--queues" in sys.argv else 0
    streams = "--streams" in sys.argv
    H, A = 128, 12
    g = torch.Generator(device=DEV).manual_seed(23)
    r = lambda *s: torch.randn(*s, generator=g, device=DEV)          # noqa: E731
    Ha, Hc = torch.nn.functional.elu(r(B, H)), torch.nn.functional.elu(r(B, H))
    Wa, ba, Wc, bc = r(A, H) / 11, r(A) * 0.1, r(1, H) / 11, r(1) * 0.1
    std = torch.rand(A, generator=g, device=DEV) + 0.5
    R = 4 * B
    actions, old_mu = r(R, A), r(R, A)
    old_sigma = torch.rand(R, A, generator=g, device=DEV) + 0.5
    old_logp, adv, ret, oldv = r(R), r(R), r(R), r(R)
    idx = torch.randperm(R, generator=g, device=DEV)[:B]
    cfg = _ffi.DtcPpoCfg()
    cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.desired_kl, cfg.use_clipped_value_loss, cfg.adaptive_schedule = 0.2, 1.0, 0.003, 0.01, 1, 0
    mean, val, dmean, dval = (torch.empty(B, w, device=DEV) for w in (A, 1, A, 1))
    dstd, losses = torch.zeros(A, device=DEV), torch.zeros(4, device=DEV)
    lr = torch.full((1,), 1e-3, dtype=torch.float64, device=DEV)
    ws = ops.workspace(_ffi.lib().dtc_loss_workspace(B), DEV)
    imgs = (h2i.HImage(B, H, DEV), h2i.HImage(B, H, DEV), h2i.HImage(B, A, DEV), h2i.HImage(B, 1, DEV))
    run = lambda: ops.ppo_heads_loss(Ha, Hc, Wa, ba, Wc, bc, "elu", std, actions, old_logp, old_mu, old_sigma, adv, ret, oldv, idx, cfg, mean, val,  # noqa: E731
                                     dmean, dval, None, None, dstd, losses, lr, ws, imgs=imgs)
    outs = lambda: [mean, val, dmean, dval, dstd, losses] + [im.buf for im in imgs]      # noqa: E731
    names = ["mean", "value", "dmean", "dvalue", "dstd", "losses", "img_dHa", "img_dHc", "img_dmean", "img_dval"]
    run()
    torch.cuda.synchronize()
    first = [t.clone() for t in outs()]
    kids = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "noise"], env=dict(os.environ, NOISE_QUEUES=str(nq))) for _ in range(n_noise)]
    if n_noise:
        time.sleep(20)                      # the children's imports
    bad = torch.zeros(len(names), dtype=torch.int64, device=DEV)
    mine = many_streams(nq)
    mine_t = [torch.randn(4096, device=DEV) for _ in mine]
    side = torch.cuda.Stream()
    junk = torch.randn(1 << 20, device=DEV)
    t0 = time.time()
    for i in range(reps):
        for t in outs()[:6]:
            t.fill_(float("nan"))
        run()
        bad += torch.stack([(a.view(torch.int32) != b.view(torch.int32)).any() if a.dtype == torch.float32 else (a != b).any()
                            for a, b in zip(outs(), first)]).to(torch.int64)
        if streams:
            with torch.cuda.stream(side):
                junk = torch.sin(junk) * 1.0001
        for s_, t_ in zip(mine, mine_t):
            with torch.cuda.stream(s_):
                t_.mul_(1.0001)
        if i % 500 == 499:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.time() - t0
    for k in kids:
        k.terminate()
    print(f"B={B} reps={reps} unroll={os.environ.get('DTC_HEADS_UNROLL', '0')} noise={n_noise} streams={streams} queues={nq}: launches whose output differed "
          f"from launch 0: " + ", ".join(f"{n}={int(c)}" for n, c in zip(names, bad.tolist())) + f"  ({dt:.0f} s)")


if __name__ == "__main__":
    main()
