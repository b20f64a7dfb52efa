"""torch-CPU restatement of BASELINE config 5, "GRU + CE-net + foothold obs" (TEST INFRASTRUCTURE ONLY).

The reference has no such model (SURVEY.md §0, §8a "Config 5"): it is the build-defined composition
  * `Vae` feature builders unchanged (actor_critic_decoder.py:286-302, 409-437, 540-551):
        actor features  = cat[obs, z, mu[:, :3], l_t]                              (584)
        critic features = cat[obs, base_vel, priv[:, 693:696], priv[:, 696:]]      (752)
  * `ActorCriticRecurrent(584, 752, 12, [512,256,128], [512,256,128], 'elu', rnn_type='gru', 512, 1)`
    (actor_critic_recurrent.py:40-116): Memory(584->512) -> actor MLP, Memory(752->512) -> critic MLP;
  * per mini-batch (N/4 envs x all 24 steps, rollout_storage.py:217-267): the VAE step of ppo.py:197-258 on the
    valid (un-padded) rows, then the PPO step of ppo.py:265-338 with BPTT over the padded trajectories, the
    policy gradient flowing through the GRU input into z, mu and l_t exactly as it does through actor_body.
Pinned by tests/golden/composite.npz: forward outputs and parameter gradients of the same composition built from
the IMPORTED reference classes (`Vae`, `ActorCriticRecurrent`, `split_and_pad_trajectories`, `unpad_trajectories`).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .gru_ref import RefActorCriticRecurrent, split_and_pad, unpad
from .ppo_ref import N_HEIGHT, RefPPO, RefVae, StepRecord

ACTOR_FEATURES, CRITIC_FEATURES, RNN_HIDDEN = 53 + 16 + 3 + 512, 53 + 3 + 3 + 693, 512


class RefCompositeAC(nn.Module):
    """`vae` (RefVae) + `acr` (RefActorCriticRecurrent on the 584 / 752-wide features): the same two sub-modules,
    in the same order, as the composition of the imported reference classes that generated the fixture."""
    is_recurrent = True

    def __init__(self, num_actions=12, hidden=(512, 256, 128)):
        super().__init__()
        self.vae = RefVae()
        self.acr = RefActorCriticRecurrent(ACTOR_FEATURES, CRITIC_FEATURES, num_actions, hidden, RNN_HIDDEN)

    actor = property(lambda self: self.acr.actor)
    critic = property(lambda self: self.acr.critic)
    std = property(lambda self: self.acr.std)
    memory_a = property(lambda self: self.acr.memory_a)
    memory_c = property(lambda self: self.acr.memory_c)

    def actor_features(self, obs, hist, priv, eps):
        mu, lv, z = self.vae.cenet_forward(hist, eps)
        l_t = self.vae.terrain_encoder(priv[:, :N_HEIGHT])
        return torch.cat((obs, z, mu[:, :3], l_t), dim=-1)

    @staticmethod
    def critic_features(obs, priv, base_vel):
        return torch.cat((obs, base_vel, priv[:, 693:696], priv[:, 696:]), dim=-1)


def recurrent_slices(st, hid_a, hid_c, num_mini_batches):
    """The recurrent mini-batches of rollout_storage.py:217-267 as index data: env slice [a, b), time-major flat
    row indices of its (t, env) samples, the slice's `dones`, and the hidden states at its trajectory starts."""
    T, N = st.dones.shape[0], st.dones.shape[1]
    mb = N // num_mini_batches
    dones = st.dones.squeeze(-1)
    lwd = torch.zeros_like(dones, dtype=torch.bool)
    lwd[1:] = dones[:-1].bool()
    lwd[0] = True
    for i in range(num_mini_batches):
        a, b = i * mb, (i + 1) * mb
        idx = (torch.arange(T).unsqueeze(1) * N + torch.arange(a, b)).reshape(-1)
        pick = lambda h: h[:, :, a:b].permute(2, 0, 1, 3)[lwd[:, a:b].permute(1, 0)].transpose(1, 0).contiguous()
        yield dict(a=a, b=b, idx=idx, dones=st.dones[:, a:b], hid_a=pick(hid_a), hid_c=pick(hid_c))


class RefCompositePPO(RefPPO):
    """RefPPO with the recurrent policy step; `vae_step(idx, eps1, rec)` is inherited unchanged."""

    def forward(self, bt, eps):
        ac, st = self.actor_critic, self.storage
        T, nmb = bt["dones"].shape[0], bt["b"] - bt["a"]
        (obs, critic_obs, priv, hist, *_rest) = st.gather(bt["idx"])
        base_vel = st.base_vel.flatten(0, 1)[bt["idx"]]
        fa = ac.actor_features(obs, hist, priv, eps).view(T, nmb, -1)
        fc = ac.critic_features(critic_obs, priv, base_vel).view(T, nmb, -1)
        pa, masks = split_and_pad(fa, bt["dones"])
        pc, _ = split_and_pad(fc, bt["dones"])
        out_a, _ = ac.memory_a.rnn(pa, bt["hid_a"])
        out_c, _ = ac.memory_c.rnn(pc, bt["hid_c"])
        mean = ac.actor(unpad(out_a, masks)).flatten(0, 1)
        value = ac.critic(unpad(out_c, masks)).flatten(0, 1)
        return mean, value

    def ppo_step(self, bt, eps2, rec: StepRecord) -> StepRecord:
        (_o, _c, _p, _h, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, *_r) = \
            self.storage.gather(bt["idx"])
        mean, value = self.forward(bt, eps2)
        return self._ppo_tail(mean, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, rec)

    def step(self, bt, eps1, eps2) -> StepRecord:
        rec = StepRecord()
        self.vae_step(bt["idx"], eps1, rec)
        self.ppo_step(bt, eps2, rec)
        return rec
