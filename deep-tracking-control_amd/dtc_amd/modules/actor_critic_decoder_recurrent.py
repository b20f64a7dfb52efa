"""ActorCriticDecoderRecurrent -- BASELINE.json config 5, "GRU + CE-net + foothold obs".

Build-defined model (the reference has none, SURVEY.md §0 / §8a "Config 5"): the `ActorCriticDecoder` feature
builders unchanged (actor_critic_decoder.py:409-437, 540-551)

    actor features  = cat[obs, z, mu[:, :3], l_t]                           (584)
    critic features = cat[obs, base_vel, priv[:, 693:696], priv[:, 696:]]   (752)

feeding the reference's recurrent head (actor_critic_recurrent.py:40-116): `Memory(584 -> 512, gru)` -> actor
MLP(512 -> 512 -> 256 -> 128 -> 12) and `Memory(752 -> 512, gru)` -> critic MLP(512 -> ... -> 1).
Parameter / `state_dict()` names: `vae.*` as ActorCriticDecoder, `std`, `actor.*`, `critic.*`, `memory_a.rnn.*`,
`memory_c.rnn.*` as ActorCriticRecurrent.  Parameters live in the same flat arena as ActorCriticDecoder (main
optimiser range = heads + shared encoders, VAE optimiser range = the whole `vae`).

Compute: CE-net / terrain encoders on the valid rows (dtc_linear_* with the mini-batch gather folded in), GRU
input projection on the valid rows (the four feature blocks are never concatenated), recurrence = dtc_gru_fwd.
Training (BPTT + the VAE step) is driven by dtc_amd.algorithms.RecurrentDecoderPPO.  GPU only.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _ffi, ops
from .actor_critic_decoder import AC_Args, ActorCriticDecoder, Dense, ParamArena, Vae, get_activation
from .actor_critic_recurrent import Memory, _mlp


class ActorCriticDecoderRecurrent(ActorCriticDecoder):
    is_recurrent = True

    def __init__(self, num_obs, num_critic_obs, num_actions, actor_hidden_dims=[512, 256, 128],
                 critic_hidden_dims=[512, 256, 128], activation='elu', rnn_type='gru', rnn_hidden_size=512,
                 rnn_num_layers=1, init_noise_std=1.0, **kwargs):
        if kwargs:
            print("ActorCriticDecoderRecurrent.__init__ got unexpected arguments, which will be ignored: "
                  + str([key for key in kwargs.keys()]))
        nn.Module.__init__(self)
        if rnn_type.lower() != 'gru' or rnn_num_layers != 1:
            raise NotImplementedError("the composite model of BASELINE.json configs[4] is built for a 1-layer GRU "
                                      "(ActorCriticRecurrent / Memory take 'lstm' and deeper stacks)")
        if activation not in ("elu", "relu"):
            raise NotImplementedError("the HIP layers implement 'elu' and 'relu'")
        A = AC_Args
        self.activation_name = activation
        self.num_obs, self.num_critic_obs, self.num_actions = num_obs, num_critic_obs, num_actions
        self.actor_features = num_obs + 16 + 3 + A.terrain_encoder_branch_latent_dims[0]
        self.critic_features = 693 + num_obs + 3 + 3
        self.rnn_hidden_size = rnn_hidden_size
        # construction order = Vae(), then ActorCriticRecurrent (actor, critic, std, memory_a, memory_c)
        self.vae = Vae()
        self.actor = _mlp(rnn_hidden_size, actor_hidden_dims, num_actions, get_activation(activation))
        self.critic = _mlp(rnn_hidden_size, critic_hidden_dims, 1, get_activation(activation))
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.memory_a = Memory(self.actor_features, type=rnn_type, num_layers=rnn_num_layers, hidden_size=rnn_hidden_size)
        self.memory_c = Memory(self.critic_features, type=rnn_type, num_layers=rnn_num_layers, hidden_size=rnn_hidden_size)
        self.distribution = None
        self.arena: ParamArena | None = None
        self._fw = {}
        self._dist = None

    def _build_layers(self, ar):
        act = self.activation_name
        self.L = self._vae_layers(ar)
        n_a = len([m for m in self.actor if isinstance(m, nn.Linear)])
        n_c = len([m for m in self.critic if isinstance(m, nn.Linear)])
        self.A = [ar.dense(f"actor.{2 * i}.weight", f"actor.{2 * i}.bias", act if i < n_a - 1 else None) for i in range(n_a)]
        self.Cr = [ar.dense(f"critic.{2 * i}.weight", f"critic.{2 * i}.bias", act if i < n_c - 1 else None) for i in range(n_c)]
        self.memory_a.bind(ar, "memory_a")
        self.memory_c.bind(ar, "memory_c")
        # the GRU input projections as dense layers (no activation): gi = X W_ih^T + b_ih
        mk = lambda m: Dense(m.W_ih, m.b_ih, m.gW_ih, m.gb_ih, None)
        self.proj_a, self.proj_c = mk(self.memory_a), mk(self.memory_c)

    # ------------------------------------------------------------------ rollout-mode forward (one env step)
    def reset(self, dones=None):
        self.memory_a.reset(dones)
        self.memory_c.reset(dones)

    def get_hidden_states(self):
        return self.memory_a.hidden_states, self.memory_c.hidden_states

    def _ensure_hidden(self, N, dev):
        for m in (self.memory_a, self.memory_c):
            if m.hidden_states is None:
                m.hidden_states = torch.zeros(1, N, self.rnn_hidden_size, device=dev)

    def _step_memory(self, mem, proj, X, N, dev):
        """One recurrent step over N envs: gi = X W_ih^T + b_ih, GRU cell, state advanced in place."""
        H = self.rnn_hidden_size
        gi = torch.empty(1, N, 3 * H, device=dev)
        ops.linear_fwd(X, proj.W, proj.b, gi.view(N, 3 * H), None, M=N)
        hs_all, gates, hn = torch.empty(2, N, H, device=dev), torch.empty(1, N, 3 * H, device=dev), torch.empty(1, N, H, device=dev)
        ws = ops.workspace(ops.gru_workspace_bytes(1, N, H), dev)
        ops.gru_fwd(gi, mem.hidden_states[0].contiguous(), mem.W_hh, mem.b_hh, hs_all, gates, hn, ws)
        mem.hidden_states = hs_all[1:2].clone()
        return mem.hidden_states[0]

    def _mlp(self, layers, X, M, dev):
        cur = X
        for L in layers:
            out = torch.empty(M, L.n_out, device=dev)
            ops.linear_fwd(cur, L.W, L.b, out, L.act, M=M)
            cur = out
        return cur

    def update_distribution(self, observations, observations_history, privileged_obs, eps=None):
        self.ensure_arena()
        obs, hist, priv = self._prep(observations), self._prep(observations_history), self._prep(privileged_obs)
        N, dev = obs.shape[0], obs.device
        ws = self._fwd_ws(N)
        if eps is None:
            eps = torch.randn(N, 16, device=dev)
        self._ensure_hidden(N, dev)
        self.cenet_forward_(ws, hist, eps)
        self.terrain_encoder_(ws, priv)
        self.latent_mu, self.latent_var, self.z = ws.mulv[:, :19], ws.mulv[:, 19:], ws.z
        h = self._step_memory(self.memory_a, self.proj_a, self.actor_input(ws, obs), N, dev)
        mean = self._mlp(self.A, h, N, dev)
        self._dist = (mean, self.std_view.detach().expand_as(mean))
        self.distribution = self._dist

    def act_inference(self, ob):
        self.update_distribution(ob["obs"], ob["obs_history"], ob["privileged_obs"],
                                 eps=torch.zeros(ob["obs"].shape[0], 16, device=ob["obs"].device))
        return self._dist[0]

    def act_expert(self, ob):
        raise NotImplementedError("the recurrent composite has no teacher/student split")

    def evaluate(self, critic_observations, privileged_observations, base_vel, **kwargs):
        self.ensure_arena()
        obs, priv, bv = self._prep(critic_observations), self._prep(privileged_observations), self._prep(base_vel)
        N, dev = obs.shape[0], obs.device
        self._ensure_hidden(N, dev)
        h = self._step_memory(self.memory_c, self.proj_c, self.critic_input(obs, bv, priv), N, dev)
        return self._mlp(self.Cr, h, N, dev)
