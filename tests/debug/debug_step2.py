"""Debug: teacher-forced flow (vae half, sync, ppo half); compare terrain-encoder backward intermediates."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deep-tracking-control_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dtc_amd import synthetic as S
from oracle.ppo_ref import StepRecord
import test_hip_ppo as T

DEV = "cuda:0"
ref, alg = T._pair(64)
perm, e1, e2 = S.update_noise(64, 24, 4, 5, seed=123)
idx = perm[:384]
rec = StepRecord()
ref.capture_grads = alg.capture_grads = True
T._sync_from_oracle(ref, alg)
ref.vae_step(idx, e1[0], rec)
alg.step_minibatch(idx, e1[0], e2[0], which="vae")
T._sync_from_oracle(ref, alg)
sd_ref, sd = ref.actor_critic.state_dict(), alg.actor_critic.state_dict()
print("max weight diff after sync:", max(float((sd[k].cpu() - v).abs().max()) for k, v in sd_ref.items()))
grads = {}
te = ref.actor_critic.vae.terrain_encoder
hooks = []
for li, nm in ((0, "z1"), (2, "z2"), (4, "lt")):
    def mk(nm):
        def fh(mod, inp, out):
            out.register_hook(lambda g, nm=nm: grads.__setitem__(nm, g.clone()))
            grads["fwd_" + nm] = out.detach().clone()
        return fh
    hooks.append(te[li].register_forward_hook(mk(nm)))
W_te2_before = sd_ref["vae.terrain_encoder.4.weight"].clone()
ref.ppo_step(idx, e2[0], rec)
for h in hooks:
    h.remove()
alg.step_minibatch(idx, e1[0], e2[0], which="ppo")
ac = alg.actor_critic
fw, tw, L = ac._fwd_ws(384), alg._train_ws(384), ac.L
def rel(a, b):
    return float((a.double().cpu() - b.double()).abs().max() / (b.abs().max() + 1e-30))
# NOTE: the LAST forward of the oracle's terrain encoder in ppo_step is the one hooked
print("lt fwd  rel", rel(fw.lt, grads["fwd_lt"]), " t2 mask mism", int(((fw.t2.cpu() > 0) != (grads["fwd_z2"] > 0)).sum()),
      " t1 mask mism", int(((fw.t1.cpu() > 0) != (grads["fwd_z1"] > 0)).sum()))
print("dlt     rel", rel(tw.dlt, grads["lt"]))
print("gA      rel", rel(tw.gA, grads["z2"]))
print("gB      rel", rel(tw.gB, grads["z1"]))
gA_self = (tw.dlt.double() @ W_te2_before.to(DEV).double()) * (fw.t2 > 0)
print("gA self-consistency with PRE-step oracle W_te2:", rel(tw.gA, gA_self.cpu()))
gA_ref_manual = (grads["lt"].double() @ W_te2_before.double()) * (grads["fwd_z2"] > 0)
print("oracle gA vs manual (pre-step W):", rel(grads["z2"], gA_ref_manual))
