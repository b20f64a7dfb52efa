"""Digest of every output of dtc_ppo_heads_loss_img on fixed inputs (B = 24576, H = 128, A = 12) + its time per call: run with and
without DTC_HEADS_UNROLL=0 to check that the unrolled form is bit-identical."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import _ffi, h2i, ops  # noqa: E402

DEV = "cuda:0"
B, H, A = 24576, 128, 12
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
adv[idx[:200]] = 0.0
cfg = _ffi.DtcPpoCfg()
cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef, cfg.desired_kl, cfg.use_clipped_value_loss, cfg.adaptive_schedule = 0.2, 1.0, 0.003, 0.01, 1, 0
mean, val, dmean, dval = (torch.empty(B, w, device=DEV) for w in (A, 1, A, 1))
dHa, dHc = torch.empty(B, H, device=DEV), torch.empty(B, H, device=DEV)
dstd, losses = torch.zeros(A, device=DEV), torch.zeros(4, device=DEV)
lr = torch.full((1,), 1e-3, dtype=torch.float64, device=DEV)
ws = ops.workspace(_ffi.lib().dtc_loss_workspace(B), DEV)
imgs = (h2i.HImage(B, H, DEV), h2i.HImage(B, H, DEV), h2i.HImage(B, A, DEV), h2i.HImage(B, 1, DEV))
run = lambda fp32: ops.ppo_heads_loss(Ha, Hc, Wa, ba, Wc, bc, "elu", std, actions, old_logp, old_mu, old_sigma, adv, ret, oldv, idx, cfg, mean, val,  # noqa: E731
                                      dmean, dval, dHa if fp32 else None, dHc if fp32 else None, dstd, losses, lr, ws, imgs=imgs)
run(True)
torch.cuda.synchronize()
h = hashlib.sha256()
for t in (mean, val, dmean, dval, dHa, dHc, dstd, losses) + tuple(im.buf for im in imgs):
    h.update(t.cpu().numpy().tobytes())
for _ in range(5):
    run(False)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    run(False)
e1.record()
torch.cuda.synchronize()
print("digest", h.hexdigest()[:16], f"; {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call (heads + finalize, images only)")
