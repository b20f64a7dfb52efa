"""Upper bounds for the small-kernel work items: the bench step of configs[1] timed with whole launch classes REMOVED (the
results are then wrong -- timing only).  What a class costs in the overlapped schedule is what the step gains when it costs nothing;
a fused / folded version of the class can gain at most that.

    python tools/skip_probe.py [steps]         -> one line per variant: ms per step (median of 3 runs of `steps` steps)

Classes: narrow_fwd / narrow_dgrad (single-pass fp32 kernels of the < 128-column layers), narrow_wgrad (their grouped weight gradients,
second side stream), latent (cenet_latent_fwd / bwd), vae_loss, reduce (DTC_WGRAD_H2I_SKIP_REDUCE=1 is not a thing: the reduce is skipped
by patching the library entry to a launch of the partial kernel only -- not possible from here, so it is NOT in the list), pack
(h2i.HImage.pack of the gathered rollout rows), adam."""
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import foothold, h2i, ops, synthetic as S  # noqa: E402
from dtc_amd.algorithms import PPO  # noqa: E402
from dtc_amd.algorithms import ppo as ppo_mod  # noqa: E402
from dtc_amd.modules import ActorCriticDecoder  # noqa: E402
from dtc_amd.modules import actor_critic_decoder as acd  # noqa: E402

DEV = "cuda:0"
N, T = 4096, 24
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 6
data = S.rollout(N, T, seed=4, device=DEV)
sc = S.scorer_inputs(N * T, seed=7, device=DEV)
last = {k: data[k][-1] for k in ("observations", "privileged_observations", "base_vel")}
torch.manual_seed(3)
ac = ActorCriticDecoder(53, 1389, 12)
alg = PPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=DEV)
alg.init_storage(N, T, [53], [1389], [265], [12])
for k, v in data.items():
    if k != "last_values":
        getattr(alg.storage, k).copy_(v)


def step():
    foothold.plan(sc["measured_heights"], sc["root_states"], sc["thigh_pos"], sc["commands"])
    alg.compute_returns(last["observations"], last["privileged_observations"], last["base_vel"])
    alg.storage.step = T
    return alg.update()


def run():
    out = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return statistics.median(out), min(out)


orig = dict(fwd=ops.linear_fwd, dgrad=ops.linear_dgrad, wgrad=ops.wgrad_group, lat_f=ops.cenet_latent_fwd, lat_b=ops.cenet_latent_bwd,
            vae=ops.vae_loss_fused, adam=ops.clip_adam, pack=h2i.HImage.pack)


def big(M):
    return M is None or M >= 8192                    # the update's mini-batches only (PPO.act / compute_returns stay real)


def fwd_skip(X, W, b, Y, act=None, M=None, mask=None, split=None):
    if split is False and big(M if M is not None else Y.shape[0]) and Y.shape[0] >= 8192:
        return Y
    return orig["fwd"](X, W, b, Y, act, M, mask, split)


def dgrad_skip(dZ, W, dX, Xsaved=None, act=None, M=None, mask=None, split=None):
    if split is False and dZ.shape[0] >= 8192:
        return
    return orig["dgrad"](dZ, W, dX, Xsaved, act, M, mask, split)


def wgrad_skip(jobs, M, workspace, stream_ptr=None, split=None):
    if split is False and M >= 8192:
        return []
    return orig["wgrad"](jobs, M, workspace, stream_ptr, split)


VARIANTS = {
    "baseline": {},
    "narrow_fwd": {"linear_fwd": fwd_skip},
    "narrow_dgrad": {"linear_dgrad": dgrad_skip},
    "narrow_wgrad": {"wgrad_group": wgrad_skip},
    "narrow_all": {"linear_fwd": fwd_skip, "linear_dgrad": dgrad_skip, "wgrad_group": wgrad_skip},
    "latent": {"cenet_latent_fwd": lambda *a, **k: None, "cenet_latent_bwd": lambda *a, **k: None},
    "vae_loss": {"vae_loss_fused": lambda *a, **k: None},
    "adam": {"clip_adam": lambda *a, **k: None},
    "narrow_all+latent+vae_loss": {"linear_fwd": fwd_skip, "linear_dgrad": dgrad_skip, "wgrad_group": wgrad_skip,
                                   "cenet_latent_fwd": lambda *a, **k: None, "cenet_latent_bwd": lambda *a, **k: None,
                                   "vae_loss_fused": lambda *a, **k: None},
    "baseline_again": {},
}

for _ in range(3):
    step()
for name, patch in VARIANTS.items():
    for k, f in patch.items():
        setattr(ops, k, f)
    for _ in range(2):
        step()
    med, lo = run()
    for k in patch:
        setattr(ops, k, orig[{"linear_fwd": "fwd", "linear_dgrad": "dgrad", "wgrad_group": "wgrad", "cenet_latent_fwd": "lat_f",
                              "cenet_latent_bwd": "lat_b", "vae_loss_fused": "vae", "clip_adam": "adam"}[k]])
    print(f"{name:32s} {med:7.2f} ms per step (min {lo:.2f})", flush=True)
