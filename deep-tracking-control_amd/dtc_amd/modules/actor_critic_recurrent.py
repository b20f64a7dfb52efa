"""ActorCritic / ActorCriticRecurrent (GRU) on the HIP kernels.

Same constructors, attribute and `state_dict()` names as rsl_rl/rsl_rl/modules/actor_critic.py:38-155 and
actor_critic_recurrent.py:40-116: `actor`, `critic` (nn.Sequential of Linear/ELU), `std`, `memory_a.rnn`,
`memory_c.rnn` (torch.nn.GRU parameter names `weight_ih_l0`, `weight_hh_l0`, `bias_ih_l0`, `bias_hh_l0`).
The reference cannot train this model through its own `PPO` at this commit; the training step
for it lives in dtc_amd.algorithms.recurrent_ppo (upstream rsl_rl PPO semantics + BPTT), BASELINE config 3.

Compute: input projection and MLPs = dtc_linear_*; recurrence = dtc_gru_fwd / dtc_gru_bwd; the un-padding of
utils.unpad_trajectories is folded into the first MLP layer as a row gather (never materialised).
Only GRU (`rnn_type='gru'`), one layer, is implemented -- that is what BASELINE.json names.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import _ffi, ops
from .._ffi import seg, segmat
from .actor_critic_decoder import Dense, get_activation


def _mlp(in_dim, hidden, out_dim, act):
    dims = [in_dim] + list(hidden) + [out_dim]
    mods = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods.append(act)
    return nn.Sequential(*mods)


class FlatArena:
    """All parameters of a module in one flat fp32 buffer (+ a gradient twin); nn.Parameter.data are views."""

    def __init__(self, model: nn.Module):
        self._named = list(model.named_parameters())
        dev = self._named[0][1].device
        total = sum(p.numel() for _, p in self._named)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.offsets, off = {}, 0
        for k, p in self._named:
            n = p.numel()
            self.flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = self.flat[off:off + n].view(p.shape)
            self.offsets[k] = (off, n, tuple(p.shape))
            off += n
        self.main_range = (0, total)

    def view(self, buf, name):
        off, n, shape = self.offsets[name]
        return buf[off:off + n].view(shape)

    def dense(self, wname, bname, act):
        return Dense(self.view(self.flat, wname), self.view(self.flat, bname), self.view(self.grad, wname),
                     self.view(self.grad, bname), act)


class ActorCritic(nn.Module):
    is_recurrent = False

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=[256, 256, 256],
                 critic_hidden_dims=[256, 256, 256], activation='elu', init_noise_std=1.0, **kwargs):
        if kwargs:
            print("ActorCritic.__init__ got unexpected arguments, which will be ignored: " + str([key for key in kwargs.keys()]))
        super().__init__()
        self.activation_name = activation
        if activation not in _ffi.ACT:
            raise NotImplementedError(f"unknown activation {activation!r} (the reference's get_activation table: "
                                      "elu, selu, relu, crelu, lrelu, tanh, sigmoid)")
        act = get_activation(activation)
        self.num_actions = num_actions
        self.actor = _mlp(num_actor_obs, actor_hidden_dims, num_actions, act)
        self.critic = _mlp(num_critic_obs, critic_hidden_dims, 1, act)
        self.std = nn.Parameter(init_noise_std * torch.ones(num_actions))
        self.distribution = None
        self.arena = None
        self._dist = None

    def _apply(self, fn, *a, **k):
        before = self.std.data_ptr()
        out = super()._apply(fn, *a, **k)
        if self.std.data_ptr() != before:        # storage really moved (see ActorCriticDecoder._apply)
            self.arena = None
        return out

    def reset(self, dones=None):
        pass

    def forward(self):
        raise NotImplementedError

    @property
    def action_mean(self):
        return self._dist[0]

    @property
    def action_std(self):
        return self._dist[1]

    @property
    def entropy(self):
        return (0.5 + 0.5 * float(np.log(2 * np.pi)) + torch.log(self._dist[1])).sum(dim=-1)

    def ensure_arena(self):
        if self.arena is None or self.std.data_ptr() != self.arena.view(self.arena.flat, "std").data_ptr():
            if not self.std.is_cuda:
                raise _ffi.DtcError("this model computes on the GPU only (there is no CPU fallback)")
            self.arena = FlatArena(self)
            ar, act = self.arena, self.activation_name
            n_a = len([m for m in self.actor if isinstance(m, nn.Linear)])
            n_c = len([m for m in self.critic if isinstance(m, nn.Linear)])
            self.A = [ar.dense(f"actor.{2 * i}.weight", f"actor.{2 * i}.bias", act if i < n_a - 1 else None) for i in range(n_a)]
            self.Cr = [ar.dense(f"critic.{2 * i}.weight", f"critic.{2 * i}.bias", act if i < n_c - 1 else None) for i in range(n_c)]
            self.std_view, self.std_grad = ar.view(ar.flat, "std"), ar.view(ar.grad, "std")
            self._build_extra(ar)
        return self.arena

    def _build_extra(self, ar):
        pass

    # -- MLP forward over M rows; returns the list of layer outputs (last = result), X may be a DtcSegMat
    @staticmethod
    def mlp_forward(layers, X, M, dev, outs=None):
        outs = outs or [torch.empty(M, L.n_out, device=dev) for L in layers]
        cur = X
        for L, o in zip(layers, outs):
            ops.linear_fwd(cur, L.W, L.b, o, L.act, M=M)
            cur = o
        return outs

    def update_distribution(self, observations):
        self.ensure_arena()
        x = observations.contiguous().float()
        mean = self.mlp_forward(self.A, x, x.shape[0], x.device)[-1]
        self._dist = (mean, self.std_view.detach().expand_as(mean))
        self.distribution = self._dist

    def act(self, observations, noise=None, **kwargs):
        self.update_distribution(observations)
        mean = self._dist[0]
        if noise is None:
            noise = torch.randn_like(mean)
        actions, self._logp = torch.empty_like(mean), torch.empty(mean.shape[0], device=mean.device)
        ops.gaussian_act(mean, self.std_view, noise, actions, self._logp)
        self._last_actions = actions
        return actions

    def get_actions_log_prob(self, actions):
        mean, sigma = self._dist
        if getattr(self, "_last_actions", None) is actions:
            return self._logp
        return (-((actions - mean) ** 2) / (2 * sigma * sigma) - torch.log(sigma) - float(np.log(np.sqrt(2 * np.pi)))).sum(dim=-1)

    def act_inference(self, observations):
        self.update_distribution(observations)
        return self._dist[0]

    def evaluate(self, critic_observations, **kwargs):
        self.ensure_arena()
        x = critic_observations.contiguous().float()
        return self.mlp_forward(self.Cr, x, x.shape[0], x.device)[-1]


class Memory(nn.Module):
    """GRU state holder (actor_critic_recurrent.py:92-116).  `self.rnn` keeps torch's parameter names."""

    def __init__(self, input_size, type='gru', num_layers=1, hidden_size=256):
        super().__init__()
        if type.lower() != 'gru' or num_layers != 1:
            raise NotImplementedError("the HIP recurrence implements a 1-layer GRU (BASELINE.json config 3)")
        self.rnn = nn.GRU(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.hidden_states = None
        self.input_size, self.hidden_size = input_size, hidden_size
        self.saved = None          # tensors of the last batch-mode forward (for BPTT)

    def bind(self, arena, prefix):
        v = lambda buf, n: arena.view(buf, f"{prefix}.rnn.{n}")
        self.W_ih, self.W_hh = v(arena.flat, "weight_ih_l0"), v(arena.flat, "weight_hh_l0")
        self.b_ih, self.b_hh = v(arena.flat, "bias_ih_l0"), v(arena.flat, "bias_hh_l0")
        self.gW_ih, self.gW_hh = v(arena.grad, "weight_ih_l0"), v(arena.grad, "weight_hh_l0")
        self.gb_ih, self.gb_hh = v(arena.grad, "bias_ih_l0"), v(arena.grad, "bias_hh_l0")

    def run(self, x, h0):
        """x [T,R,I], h0 [R,H] -> dict(hs_all [T+1,R,H], gates, hn, gi, x)."""
        T, R, I = x.shape
        H, dev = self.hidden_size, x.device
        x2 = x.contiguous().view(T * R, I)
        gi = torch.empty(T, R, 3 * H, device=dev)
        ops.linear_fwd(x2, self.W_ih, self.b_ih, gi.view(T * R, 3 * H), None)
        hs_all = torch.empty(T + 1, R, H, device=dev)
        gates, hn = torch.empty(T, R, 3 * H, device=dev), torch.empty(T, R, H, device=dev)
        ws = ops.workspace(ops.gru_workspace_bytes(T, R, H), dev)
        ops.gru_fwd(gi, h0.contiguous(), self.W_hh, self.b_hh, hs_all, gates, hn, ws)
        return dict(hs_all=hs_all, gates=gates, hn=hn, x2=x2, ws=ws, T=T, R=R)

    def backward(self, saved, dhs, wgrad=None):
        """BPTT: dhs [T,R,H] -> parameter gradients into the arena; returns dgi [T,R,3H] (gradient of the input
        projection's output).  `wgrad(dZ, X, gW, gb)` optionally takes over the input-projection weight gradient
        (the trainer runs it on its weight-gradient stream)."""
        T, R, H = saved["T"], saved["R"], self.hidden_size
        dev = dhs.device
        dgi = torch.empty(T, R, 3 * H, device=dev)
        dh0 = torch.empty(R, H, device=dev)
        ops.gru_bwd(dhs.contiguous(), saved["hs_all"], saved["gates"], saved["hn"], self.W_hh, dgi, self.gW_hh, self.gb_hh,
                    dh0, saved["ws"])
        if wgrad is not None:
            wgrad(dgi.view(T * R, 3 * H), saved["x2"], self.gW_ih, self.gb_ih)
        else:
            wws = ops.workspace(ops.wgrad_workspace_bytes(T * R, 3 * H, self.input_size), dev)
            ops.linear_wgrad(dgi.view(T * R, 3 * H), saved["x2"], self.gW_ih, self.gb_ih, wws)
        return dgi

    def forward(self, input, masks=None, hidden_states=None):
        batch_mode = masks is not None
        if batch_mode:
            if hidden_states is None:
                raise ValueError("Hidden states not passed to memory module during policy update")
            self.saved = self.run(input, hidden_states[0] if hidden_states.dim() == 3 else hidden_states)
            return self.saved["hs_all"][1:]
        if self.hidden_states is None:
            self.hidden_states = torch.zeros(1, input.shape[0], self.hidden_size, device=input.device)
        out = self.run(input.unsqueeze(0), self.hidden_states[0])
        self.hidden_states = out["hs_all"][1:2].clone()
        return self.hidden_states

    def reset(self, dones=None):
        if self.hidden_states is not None:
            self.hidden_states[..., dones.bool() if dones.dtype != torch.bool else dones, :] = 0.0


class ActorCriticRecurrent(ActorCritic):
    is_recurrent = True

    def __init__(self, num_actor_obs, num_critic_obs, num_actions, actor_hidden_dims=[256, 256, 256],
                 critic_hidden_dims=[256, 256, 256], activation='elu', rnn_type='lstm', rnn_hidden_size=256,
                 rnn_num_layers=1, init_noise_std=1.0, **kwargs):
        if kwargs:
            print("ActorCriticRecurrent.__init__ got unexpected arguments, which will be ignored: " + str(kwargs.keys()))
        super().__init__(num_actor_obs=rnn_hidden_size, num_critic_obs=rnn_hidden_size, num_actions=num_actions,
                         actor_hidden_dims=actor_hidden_dims, critic_hidden_dims=critic_hidden_dims,
                         activation=activation, init_noise_std=init_noise_std)
        self.memory_a = Memory(num_actor_obs, type=rnn_type, num_layers=rnn_num_layers, hidden_size=rnn_hidden_size)
        self.memory_c = Memory(num_critic_obs, type=rnn_type, num_layers=rnn_num_layers, hidden_size=rnn_hidden_size)
        self.rnn_hidden_size = rnn_hidden_size

    def _build_extra(self, ar):
        self.memory_a.bind(ar, "memory_a")
        self.memory_c.bind(ar, "memory_c")

    def reset(self, dones=None):
        self.memory_a.reset(dones)
        self.memory_c.reset(dones)

    def _through(self, memory, layers, observations, masks, hidden_states, unpad_idx):
        """memory -> (un-pad as a row gather) -> MLP.  Returns the MLP layer outputs."""
        self.ensure_arena()
        out = memory(observations.float(), masks, hidden_states)          # [T,R,H] or [1,N,H]
        T, R, H = out.shape
        flat = out.reshape(T * R, H)
        if masks is None:
            return self.mlp_forward(layers, flat, T * R, flat.device), flat
        X = segmat([seg(flat, 0, H, gather=True)], unpad_idx)
        return self.mlp_forward(layers, X, unpad_idx.numel(), flat.device), flat

    def act(self, observations, masks=None, hidden_states=None, unpad_idx=None, noise=None):
        outs, _ = self._through(self.memory_a, self.A, observations, masks, hidden_states, unpad_idx)
        self._actor_outs = outs
        mean = outs[-1]
        self._dist = (mean, self.std_view.detach().expand_as(mean))
        if noise is None:
            noise = torch.randn_like(mean)
        actions, self._logp = torch.empty_like(mean), torch.empty(mean.shape[0], device=mean.device)
        ops.gaussian_act(mean, self.std_view, noise, actions, self._logp)
        self._last_actions = actions
        return actions

    def act_inference(self, observations):
        outs, _ = self._through(self.memory_a, self.A, observations, None, None, None)
        return outs[-1]

    def evaluate(self, critic_observations, masks=None, hidden_states=None, unpad_idx=None):
        outs, _ = self._through(self.memory_c, self.Cr, critic_observations, masks, hidden_states, unpad_idx)
        self._critic_outs = outs
        return outs[-1]

    def get_hidden_states(self):
        return self.memory_a.hidden_states, self.memory_c.hidden_states
