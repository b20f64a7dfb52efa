"""PPO for the recurrent (GRU / LSTM) actor-critic -- BASELINE.json config 3 ("ActorCriticRecurrent (GRU hidden
512) BPTT over 24 steps").

The reference's `PPO` cannot drive `ActorCriticRecurrent` at this commit (it needs `.vae`, ppo.py:79, and its
update unpacks 16 items where the recurrent generator yields 11 -- SURVEY.md F2), so this class is the upstream
rsl_rl PPO step the code base was forked from: ppo.py:288-335 (log-prob / entropy / KL-adaptive learning rate /
clipped surrogate + clipped value loss / clip_grad_norm_ / Adam) fed by `reccurent_mini_batch_generator`
(rollout_storage.py:217-267), with hidden states recorded BEFORE each rollout step (the convention of the
commented lines ppo.py:138-139).  Same constructor keywords and method names as `PPO`.

Kernel schedule per mini-batch (N/4 envs x all 24 steps): input projection GEMM -> dtc_gru_fwd -> MLP with the
un-padding folded in as a row gather -> fused PPO loss -> MLP backward -> row scatter -> dtc_gru_bwd (BPTT) ->
input-projection weight gradient -> one fused clip+Adam over the flat parameter arena.
"""
from __future__ import annotations

import torch

from .. import _ffi, distributed as dp, ops
from .._ffi import seg, segmat
from ..modules.actor_critic_recurrent import ActorCriticRecurrent
from ..storage import RolloutStorage
from ..utils import true_indices
import os

from .ppo import FusedAdam, S_ENTROPY, S_GNORM, S_KL, S_SURR, S_VALUE, STAT_COLS, _Lanes


class RecurrentPPO:
    actor_critic: ActorCriticRecurrent

    def __init__(self, actor_critic, num_learning_epochs=5, num_mini_batches=4, clip_param=0.2, gamma=0.99, lam=0.95,
                 value_loss_coef=1.0, entropy_coef=0.01, learning_rate=5.e-4, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01, device='cpu'):
        self.device = device
        self.desired_kl, self.schedule, self.learning_rate = desired_kl, schedule, learning_rate
        self.actor_critic = actor_critic
        self.actor_critic.to(self.device)
        self.storage = None
        self.optimizer = None
        if torch.device(device).type == "cuda":
            arena = actor_critic.ensure_arena()
            self.optimizer = FusedAdam(arena, arena.main_range, actor_critic.parameters(), lr=learning_rate)
            from .. import distributed as dp
            dp.broadcast_parameters_(arena.flat)             # data parallel: all ranks start from rank 0's weights
        self.transition = RolloutStorage.Transition()
        self.clip_param, self.num_learning_epochs, self.num_mini_batches = clip_param, num_learning_epochs, num_mini_batches
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.gamma, self.lam, self.max_grad_norm = gamma, lam, max_grad_norm
        self.use_clipped_value_loss = use_clipped_value_loss
        self.capture_grads, self.captured = False, {}
        self.last_update_stats = None
        # actor and critic are independent recurrences until the loss and again until the optimiser step: the critic
        # runs on a second stream, every weight gradient on a third (DTC_OVERLAP_LANES=0 / DTC_OVERLAP_WGRAD=0: serial)
        self.overlap = os.environ.get("DTC_OVERLAP_LANES", "1") != "0" and os.environ.get("DTC_OVERLAP_WGRAD", "1") != "0"
        self._lanes = None
        self._wimages = None

    def _require_gpu(self):
        if self.optimizer is None:
            raise _ffi.DtcError("dtc_amd.RecurrentPPO computes on an MI355X only (device='cuda:N')")

    def init_storage(self, num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, action_shape):
        self.storage = RolloutStorage(num_envs, num_transitions_per_env, actor_obs_shape, critic_obs_shape, [1],
                                      action_shape, self.device)

    def test_mode(self):
        self.actor_critic.eval()

    def train_mode(self):
        self.actor_critic.train()

    # ---------------------------------------------------------------- rollout side
    def act(self, obs, critic_obs):
        self._require_gpu()
        ac, tr = self.actor_critic, self.transition
        N = obs.shape[0]
        for m in (ac.memory_a, ac.memory_c):
            if m.hidden_states is None:
                m.hidden_states = m.init_hidden(N, obs.device)
        tr.hidden_states = tuple(m.clone_hidden(h) for m, h in zip((ac.memory_a, ac.memory_c), ac.get_hidden_states()))   # BEFORE this step
        tr.actions = ac.act(obs).detach()
        tr.values = ac.evaluate(critic_obs).detach()
        tr.actions_log_prob = ac.get_actions_log_prob(tr.actions).detach()
        tr.action_mean, tr.action_sigma = ac.action_mean.detach(), ac.action_std.detach()
        tr.observations, tr.critic_observations, tr.privileged_observations = obs, critic_obs, critic_obs
        tr.observation_histories = torch.zeros(N, 1, device=obs.device)
        tr.base_vel = torch.zeros(N, 3, device=obs.device)
        return tr.actions

    def process_env_step(self, rewards, dones, infos, next_obs=None):
        tr = self.transition
        tr.rewards, tr.dones = rewards.clone(), dones
        tr.next_observations = next_obs if next_obs is not None else tr.observations
        if 'time_outs' in infos:
            tr.rewards += self.gamma * torch.squeeze(tr.values * infos['time_outs'].unsqueeze(1).to(self.device), 1)
        self.storage.add_transitions(tr)
        tr.clear()
        self.actor_critic.reset(dones)

    def compute_returns(self, last_critic_obs):
        self._require_gpu()
        ac = self.actor_critic
        keep = ac.memory_c.clone_hidden(ac.memory_c.hidden_states)
        last_values = ac.evaluate(last_critic_obs).detach()
        ac.memory_c.hidden_states = keep            # the bootstrap value must not advance the critic's state
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    # ---------------------------------------------------------------- update
    def _loss_cfg(self):
        cfg = _ffi.DtcPpoCfg()
        cfg.clip_param, cfg.value_loss_coef, cfg.entropy_coef = self.clip_param, self.value_loss_coef, self.entropy_coef
        cfg.desired_kl = float(self.desired_kl) if self.desired_kl is not None else 0.0
        cfg.use_clipped_value_loss = int(bool(self.use_clipped_value_loss))
        cfg.adaptive_schedule = int(self.desired_kl is not None and self.schedule == 'adaptive' and dp.world_size() == 1)
        return cfg

    def _wgrad(self, ln, dZ, X, gW, gb, M, rows=None):
        """Weight gradient on the side stream (off the critical path until the optimiser step)."""
        N, K = gW.shape
        need = ops.wgrad_workspace_bytes(M, N, K)
        if ln.wg is None or ln.wg.numel() * ln.wg.element_size() < need:
            torch.cuda.synchronize()
            ln.wg = ops.workspace(need, gW.device)
        if self.overlap:
            ev = ln.event()
            ev.record()
            ln.side.wait_event(ev)
            ops.linear_wgrad(dZ, X, gW, gb, ln.wg, M=M, stream_ptr=ln.side.cuda_stream, rows=rows)
            ln.side_busy = True
        else:
            ops.linear_wgrad(dZ, X, gW, gb, ln.wg, M=M, rows=rows)

    def _mlp_backward(self, ln, layers, outs, dOut, X0, M, dev, keep):
        """Backward through an MLP given the saved layer outputs; returns the gradient w.r.t. its input rows.
        Every gradient buffer goes into `keep`: the side stream still reads it for the weight gradient after this
        lane has moved on, so it must not return to the caching allocator before the join."""
        dZ = dOut
        for li in range(len(layers) - 1, -1, -1):
            L = layers[li]
            X = outs[li - 1] if li > 0 else X0
            self._wgrad(ln, dZ, X, L.gW, L.gb, M)
            dX = torch.empty(M, L.n_in, device=dev)
            keep.append(dX)
            if li > 0:
                ops.linear_dgrad(dZ, L.W, dX, outs[li - 1], layers[li - 1].act, M=M)
            else:
                ops.linear_dgrad(dZ, L.W, dX, None, None, M=M)
            dZ = dX
        return dZ

    def step_minibatch(self, batch, start, stop, stats=None):
        """One recurrent mini-batch: `batch` = 11-tuple of reccurent_mini_batch_generator, envs [start, stop)."""
        self._require_gpu()
        ac, st = self.actor_critic, self.storage
        arena = ac.ensure_arena()
        if self.optimizer.arena is not arena:            # the model moved: re-bind the optimiser's views
            arena._named = list(ac.named_parameters())
            self.optimizer.rebind(arena)
        (obs_b, cobs_b, _a, _v, _adv, _r, _lp, _mu, _sg, (hid_a, hid_c), masks) = batch
        dev = obs_b.device
        T, R = masks.shape
        N, Nmb = st.num_envs, stop - start
        M = T * Nmb
        if self._lanes is None:
            self._lanes = _Lanes(dev)
            self._lanes.wg = None
        ln = self._lanes
        # un-padding as a row map: padded row (pos*R + traj) of each (t, env) in time-major order
        flat_rt = true_indices(masks.transpose(1, 0), M)          # every (env, t) has exactly one padded slot: no nonzero() sync
        traj, pos = flat_rt // T, flat_rt % T
        unpad_idx = (pos * R + traj).view(Nmb, T).transpose(1, 0).reshape(-1).contiguous()
        store_idx = (torch.arange(T, device=dev).unsqueeze(1) * N + torch.arange(start, stop, device=dev)).reshape(-1).contiguous()
        stats = torch.zeros(STAT_COLS, device=dev) if stats is None else stats
        self.optimizer.set_lr(self.learning_rate)
        _ffi.lib().dtc_set_concurrency_hint(int(bool(self.overlap)))
        if self._wimages is None:
            self._wimages = ops.WeightImages()
        with self._wimages:                          # weight images of the step's split-path layers: one launch
            self._forward_backward(ln, batch, stats, unpad_idx, store_idx, M, T, R, dev)
        arena = ac.arena
        dp_adaptive = dp.world_size() > 1 and self.desired_kl is not None and self.schedule == 'adaptive'
        if dp.world_size() > 1:
            dp.allreduce_mean_(arena.grad_full)      # header (KL) + every gradient: one collective per optimiser step
        if dp_adaptive:
            stats[S_KL:S_KL + 1].copy_(arena.kl_slot)
            ops.lr_adapt(arena.kl_slot, self.optimizer.lr_dev, float(self.desired_kl))
        if self.capture_grads:
            self.captured["main"] = ac.arena.grad.clone()
        self.optimizer.step(self.max_grad_norm, stats[S_GNORM:S_GNORM + 1])
        return stats

    def _forward_backward(self, ln, batch, stats, unpad_idx, store_idx, M, T, R, dev):
        ac, st = self.actor_critic, self.storage
        arena = ac.arena
        (obs_b, cobs_b, _a, _v, _adv, _r, _lp, _mu, _sg, (hid_a, hid_c), masks) = batch
        ln.begin(self.overlap)
        # forward: critic recurrence on the second lane
        with ln.lane("aux"):
            ac.evaluate(cobs_b, masks, hid_c, unpad_idx)
            c_outs, c_saved = ac._critic_outs, ac.memory_c.saved
        ac.act(obs_b, masks, hid_a, unpad_idx)
        a_outs, a_saved = ac._actor_outs, ac.memory_a.saved
        ln.order("aux", "main")
        mean, value = a_outs[-1], c_outs[-1]
        # loss (rows of the rollout tensors are addressed through store_idx -- no slicing copies)
        dmean, dval = torch.empty_like(mean), torch.empty(M, 1, device=dev)
        lws = ops.workspace(_ffi.lib().dtc_loss_workspace(M), dev)
        flat = lambda k: st.flat(k)
        ops.ppo_loss(mean, ac.std_view, value, flat("actions"), flat("actions_log_prob"), flat("mu"), flat("sigma"),
                     flat("advantages"), flat("returns"), flat("values"), store_idx, self._loss_cfg(), dmean, dval,
                     ac.std_grad, stats[S_SURR:S_SURR + 4], self.optimizer.lr_dev, lws)
        dp_adaptive = dp.world_size() > 1 and self.desired_kl is not None and self.schedule == 'adaptive'
        if dp_adaptive:                              # the KL mean travels in the header of the gradient exchange
            arena.kl_slot.copy_(stats[S_KL:S_KL + 1])
        ln.order("main", "aux")
        # backward: MLPs -> scatter into the padded layout -> BPTT -> input-projection weight gradient
        H = ac.rnn_hidden_size
        keep = []                                   # buffers read by the side stream stay alive until the join

        def head_backward(layers, outs, saved, mem, dOut):
            hs_flat = saved["hs_all"][1:].reshape(T * R, H)
            X0 = segmat([seg(hs_flat, 0, H, gather=True)], unpad_idx)
            d_in = self._mlp_backward(ln, layers, outs, dOut, X0, M, dev, keep)
            dhs = torch.zeros(T * R, H, device=dev)
            ops.scatter_rows(d_in, unpad_idx, dhs)
            # unpad_idx doubles as the list of valid (t, r) slots: the recurrent weight gradients skip the padding
            dgi = mem.backward(saved, dhs.view(T, R, H), rows=unpad_idx,
                               wgrad=lambda dZ, X, gW, gb: self._wgrad(ln, dZ, X, gW, gb, T * R, rows=unpad_idx))
            keep.extend((d_in, dhs, dgi, X0, outs, saved))

        with ln.lane("aux"):
            head_backward(ac.Cr, c_outs, c_saved, ac.memory_c, dval)
        head_backward(ac.A, a_outs, a_saved, ac.memory_a, dmean)
        ln.join()

    def update(self):
        self._require_gpu()
        st = self.storage
        nmb, epochs = self.num_mini_batches, self.num_learning_epochs
        mb = st.num_envs // nmb
        dev = self.actor_critic.std.device
        stats = torch.zeros(nmb * epochs, STAT_COLS, device=dev)
        k = 0
        for mem in (self.actor_critic.memory_a, self.actor_critic.memory_c):
            mem.new_update()
        for batch in st.reccurent_mini_batch_generator(nmb, epochs):
            i = k % nmb
            self.step_minibatch(batch, i * mb, (i + 1) * mb, stats[k])
            k += 1
        host = stats.cpu()
        self.learning_rate = float(self.optimizer.lr_dev.item())
        self.last_update_stats = host
        m = host.double().mean(dim=0)
        st.clear()
        return float(m[S_VALUE]), float(m[S_SURR])
