"""Debug aid: full-size composite VAE step, HIP gradients under two split-threshold settings against each other and the oracle."""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path[:0] = [R, os.path.join(R, "deep-tracking-control_amd"), os.path.join(R, "tests")]
import torch
import test_composite_path as TC
from oracle import ppo_ref as OP
from oracle import composite_ref as CR
from dtc_amd import ops
from dtc_amd.algorithms import ppo as P

n = 4096
data, hid_a, hid_c, eps, _, _ = TC.composite_case(n=n)
ref = TC.oracle_alg(data, n); ref.capture_grads = True
bt_ref = next(iter(CR.recurrent_slices(ref.storage, hid_a, hid_c, TC.NMB)))
res = {}
for cols, red in ((256, 384), (128, 128), (128, 384), (256, 128)):
    ops.SPLIT_MIN_COLS, ops.SPLIT_MIN_RED = cols, red
    alg = TC._hip_alg(ref, data, n); alg.capture_grads = True
    bt = next(iter(alg.recurrent_slices(hid_a.to(TC.DEV), hid_c.to(TC.DEV))))
    row = alg.step_minibatch(bt, eps[0].to(TC.DEV), eps[0].to(TC.DEV), which="vae").cpu()
    res[(cols, red)] = (alg.captured["vae"].clone(), alg.actor_critic.arena, row)
    print(cols, red, "vae_gnorm", float(row[P.S_VAE_GNORM]), "recons", float(row[P.S_RECONS]), "height", float(row[P.S_HEIGHT]))
rec = OP.StepRecord()
ref.vae_step(bt_ref["idx"], eps[0], rec)
print("oracle vae_gnorm", rec.vae_gnorm)
for key, (g, arena, row) in res.items():
    print("==", key)
    out = []
    for name, g_ref in rec.extra["vae_grads"].items():
        name = name.replace("acr.", "")
        v = arena.view(g, name).cpu()
        out.append((float((v - g_ref).norm() / (g_ref.norm() + 1e-30)), name, float(v.norm()), float(g_ref.norm())))
    out.sort(reverse=True)
    for o in out[:8]:
        print("   ", o)
