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
    HEADER = 4

    @property
    def kl_slot(self):
        return self.grad_full[0:1]

    def __init__(self, model: nn.Module):
        self._named = list(model.named_parameters())
        dev = self._named[0][1].device
        total = sum(p.numel() for _, p in self._named)
        self.flat = torch.empty(total, dtype=torch.float32, device=dev)
        # [kl, 0, 0, 0 | gradients]: the data-parallel exchange sends header + gradients as one all-reduce (see ParamArena)
        self.grad_full = torch.zeros(self.HEADER + total, dtype=torch.float32, device=dev)
        self.grad = self.grad_full[self.HEADER:]
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
    """Recurrent state holder (actor_critic_recurrent.py:92-116): torch.nn.GRU or torch.nn.LSTM (the reference's default),
    any number of layers.  `self.rnn` keeps torch's parameter names; the recurrence runs in csrc/gru.hip / csrc/lstm.hip.
    `hidden_states` follows torch: a [num_layers, N, H] tensor (GRU) or a tuple (h, c) of two such tensors (LSTM)."""

    def __init__(self, input_size, type='lstm', num_layers=1, hidden_size=256):
        super().__init__()
        kind = type.lower()
        if kind not in ('gru', 'lstm') or num_layers < 1:
            raise ValueError(f"Memory: type must be 'gru' or 'lstm' and num_layers >= 1 (got {type!r}, {num_layers})")
        rnn_cls = nn.GRU if kind == 'gru' else nn.LSTM
        self.rnn = rnn_cls(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)
        self.kind, self.num_layers, self.G = kind, num_layers, (3 if kind == 'gru' else 4)
        self.hidden_states = None
        self.input_size, self.hidden_size = input_size, hidden_size
        self.saved = None          # tensors of the last batch-mode forward (for BPTT)

    def bind(self, arena, prefix):
        v = lambda buf, n: arena.view(buf, f"{prefix}.rnn.{n}")
        L = range(self.num_layers)
        self.Wih, self.Whh = [v(arena.flat, f"weight_ih_l{l}") for l in L], [v(arena.flat, f"weight_hh_l{l}") for l in L]
        self.bih, self.bhh = [v(arena.flat, f"bias_ih_l{l}") for l in L], [v(arena.flat, f"bias_hh_l{l}") for l in L]
        self.gWih, self.gWhh = [v(arena.grad, f"weight_ih_l{l}") for l in L], [v(arena.grad, f"weight_hh_l{l}") for l in L]
        self.gbih, self.gbhh = [v(arena.grad, f"bias_ih_l{l}") for l in L], [v(arena.grad, f"bias_hh_l{l}") for l in L]
        # layer 0 under the names the composite model (1-layer GRU) uses
        self.W_ih, self.W_hh, self.b_ih, self.b_hh = self.Wih[0], self.Whh[0], self.bih[0], self.bhh[0]
        self.gW_ih, self.gW_hh, self.gb_ih, self.gb_hh = self.gWih[0], self.gWhh[0], self.gbih[0], self.gbhh[0]

    # ---- hidden-state helpers (GRU: tensor, LSTM: (h, c))
    def init_hidden(self, N, device):
        z = lambda: torch.zeros(self.num_layers, N, self.hidden_size, device=device)
        return z() if self.kind == 'gru' else (z(), z())

    @staticmethod
    def clone_hidden(h):
        if h is None:
            return None
        return tuple(t.clone() for t in h) if isinstance(h, (tuple, list)) else h.clone()

    def _split(self, hidden):
        """-> (h [L,R,H], c [L,R,H] or None)"""
        if self.kind == 'lstm':
            if not isinstance(hidden, (tuple, list)) or len(hidden) != 2:
                raise ValueError("LSTM memory needs hidden states (h, c)")
            h, c = hidden
        else:
            h, c = (hidden[0] if isinstance(hidden, (tuple, list)) else hidden), None
        fix = lambda t: t if t.dim() == 3 else t.unsqueeze(0)
        return fix(h), (fix(c) if c is not None else None)

    def new_update(self):
        """Called by the trainers at the start of an update: the padding slots of the compacted input projection start from zero, so
        an update's result never depends on what earlier updates left there (within an update they hold finite values of its own
        earlier mini-batches, which nothing reads: the outputs of padding steps are masked out and their gradients are zero)."""
        buf = getattr(self, "_gi_pad", None)
        if buf is not None:
            buf.zero_()

    def _padded_gi(self, rows_total, width, dev):
        """Persistent, zero-initialised [T * R, G * H] buffer for the compacted input projection (grown on demand): the slots a
        mini-batch does not write keep finite values of earlier mini-batches of the same update (new_update)."""
        buf = getattr(self, "_gi_pad", None)
        if buf is None or buf.numel() < rows_total * width or buf.device != dev:
            buf = self._gi_pad = torch.zeros(rows_total * width, device=dev)
        return buf[:rows_total * width].view(rows_total, width)

    def run(self, x, hidden, rows=None):
        """x [T,R,I], hidden as in `hidden_states` ([L,R,H] or (h, c)) -> saved dict; saved['out'] = top layer's [T,R,H].
        `rows`: see forward()."""
        T, R, _ = x.shape
        H, G, dev = self.hidden_size, self.G, x.device
        h0, c0 = self._split(hidden)
        layers, cur = [], x.contiguous().view(T * R, -1)
        compact = rows is not None and ops.WGRAD_ROWS and self.kind == 'gru' and 1024 <= rows.numel() < T * R
        for l in range(self.num_layers):
            if compact and l == 0:
                # ~30 % of a padded recurrent mini-batch is padding: project the valid rows (gathered) and scatter them into place
                gi_c = torch.empty(rows.numel(), G * H, device=dev)
                ops.linear_fwd(segmat([seg(cur, 0, cur.shape[1], gather=True)], rows), self.Wih[l], self.bih[l], gi_c, None, M=rows.numel())
                gi = self._padded_gi(T * R, G * H, dev).view(T, R, G * H)
                ops.scatter_rows(gi_c, rows, gi.view(T * R, G * H))
            else:
                gi = torch.empty(T, R, G * H, device=dev)
                ops.linear_fwd(cur, self.Wih[l], self.bih[l], gi.view(T * R, G * H), None)
            hs_all = torch.empty(T + 1, R, H, device=dev)
            gates = torch.empty(T, R, G * H, device=dev)
            rec = dict(x2=cur, hs_all=hs_all, gates=gates)
            if self.kind == 'gru':
                rec["hn"] = torch.empty(T, R, H, device=dev)
                rec["ws"] = ops.workspace(ops.gru_workspace_bytes(T, R, H), dev)
                ops.gru_fwd(gi, h0[l].contiguous(), self.Whh[l], self.bhh[l], hs_all, gates, rec["hn"], rec["ws"])
            else:
                rec["cs_all"] = torch.empty(T + 1, R, H, device=dev)
                rec["ws"] = ops.workspace(ops.lstm_workspace_bytes(T, R, H), dev)
                ops.lstm_fwd(gi, h0[l].contiguous(), c0[l].contiguous(), self.Whh[l], self.bhh[l], hs_all, rec["cs_all"], gates,
                             rec["ws"])
            layers.append(rec)
            cur = hs_all[1:].reshape(T * R, H)
        top = layers[-1]
        # layer-0 records under the flat names older callers use (1-layer GRU)
        return dict(layers=layers, out=top["hs_all"][1:], hs_all=top["hs_all"], gates=top["gates"], hn=top.get("hn"),
                    x2=layers[0]["x2"], ws=top["ws"], T=T, R=R)

    def final_hidden(self, saved):
        h = torch.stack([rec["hs_all"][-1] for rec in saved["layers"]])
        if self.kind == 'gru':
            return h
        return h, torch.stack([rec["cs_all"][-1] for rec in saved["layers"]])

    def backward(self, saved, dhs, wgrad=None, rows=None):
        """BPTT: dhs [T,R,H] (gradient w.r.t. the top layer's outputs) -> parameter gradients into the arena; returns dgi
        [T,R,G*H] of the BOTTOM layer (gradient of the input projection's output).  `wgrad(dZ, X, gW, gb)` optionally takes
        over the input-projection weight gradients (the trainer runs them on its weight-gradient stream).  `rows`: the valid
        (t, r) slots of the padded batch (t * R + r, int64 device index; GRU only): weight gradients skip the padding slots."""
        T, R, H, G = saved["T"], saved["R"], self.hidden_size, self.G
        dev = dhs.device
        d_out = dhs.contiguous()
        keep = []
        for l in range(self.num_layers - 1, -1, -1):
            rec = saved["layers"][l]
            dgi = torch.empty(T, R, G * H, device=dev)
            dh0 = torch.empty(R, H, device=dev)
            if self.kind == 'gru':
                ops.gru_bwd(d_out, rec["hs_all"], rec["gates"], rec["hn"], self.Whh[l], dgi, self.gWhh[l], self.gbhh[l], dh0,
                            rec["ws"], rows=rows)
            else:
                dc0 = torch.empty(R, H, device=dev)
                ops.lstm_bwd(d_out, rec["hs_all"], rec["cs_all"], rec["gates"], self.Whh[l], dgi, self.gWhh[l], self.gbhh[l], dh0,
                             dc0, rec["ws"])
            I = self.Wih[l].shape[1]
            if wgrad is not None:
                wgrad(dgi.view(T * R, G * H), rec["x2"], self.gWih[l], self.gbih[l])
            else:
                wws = ops.workspace(ops.wgrad_workspace_bytes(T * R, G * H, I), dev)
                ops.linear_wgrad(dgi.view(T * R, G * H), rec["x2"], self.gWih[l], self.gbih[l], wws, rows=rows)
                keep.append(wws)
            if l > 0:                              # gradient w.r.t. this layer's input = the outputs of the layer below
                d_in = torch.empty(T * R, I, device=dev)
                ops.linear_dgrad(dgi.view(T * R, G * H), self.Wih[l], d_in, None, None, M=T * R)
                d_out = d_in.view(T, R, H)
            keep.append(dgi)
        saved["_bwd_keep"] = keep                  # read by side-stream weight gradients until the trainer's join
        return dgi

    def forward(self, input, masks=None, hidden_states=None, rows=None):
        """`rows` (batch mode, optional): the valid (t, r) slots t * R + r of the padded batch -- the input projection is then
        computed for those rows only (the padding slots keep finite stale values nobody reads)."""
        batch_mode = masks is not None
        if batch_mode:
            if hidden_states is None:
                raise ValueError("Hidden states not passed to memory module during policy update")
            self.saved = self.run(input, hidden_states, rows=rows)
            return self.saved["out"]
        if self.hidden_states is None:
            self.hidden_states = self.init_hidden(input.shape[0], input.device)
        out = self.run(input.unsqueeze(0), self.hidden_states)
        self.hidden_states = self.final_hidden(out)           # torch.stack: fresh tensors
        return out["out"]

    def reset(self, dones=None):
        if self.hidden_states is None:
            return
        states = self.hidden_states if isinstance(self.hidden_states, (tuple, list)) else (self.hidden_states,)
        if dones is None:                 # actor_critic_recurrent.py:112-116 with dones=None: `t[..., None, :] = 0` clears every env
            for t in states:
                t.zero_()
            return
        d = dones.bool() if dones.dtype != torch.bool else dones
        for t in states:
            t[..., d, :] = 0.0


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
        out = memory(observations.float(), masks, hidden_states, rows=unpad_idx if masks is not None else None)   # [T,R,H] or [1,N,H]
        T, R, H = out.shape
        flat = out.reshape(T * R, H)
        if masks is None:
            return self.mlp_forward(layers, flat, T * R, flat.device), flat
        X = segmat([seg(flat, 0, H, gather=True)], unpad_idx)
        return self.mlp_forward(layers, X, unpad_idx.numel(), flat.device), flat

    def act(self, observations, masks=None, hidden_states=None, unpad_idx=None, noise=None):
        self.ensure_arena()                       # self.A / self.Cr exist from here on (first call on a fresh model)
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
        self.ensure_arena()
        outs, _ = self._through(self.memory_a, self.A, observations, None, None, None)
        return outs[-1]

    def evaluate(self, critic_observations, masks=None, hidden_states=None, unpad_idx=None):
        self.ensure_arena()
        outs, _ = self._through(self.memory_c, self.Cr, critic_observations, masks, hidden_states, unpad_idx)
        self._critic_outs = outs
        return outs[-1]

    def get_hidden_states(self):
        return self.memory_a.hidden_states, self.memory_c.hidden_states
