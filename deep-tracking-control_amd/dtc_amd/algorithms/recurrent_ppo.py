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

from .. import _ffi, distributed as dp, h2i, ops
from .._ffi import seg, segmat
from ..modules.actor_critic_recurrent import ActorCriticRecurrent
from ..storage import RolloutStorage
from ..utils import true_indices
import os

from .ppo import FusedAdam, S_ENTROPY, S_GNORM, S_KL, S_SURR, S_VALUE, STAT_COLS, _Lanes


def _share_rule():
    """Several ranks of a job on ONE device (the gloo rehearsals of the data-parallel path): more than two persistent recurrence launches
    could be in flight on the device at once, and their workgroups must all be resident to meet (csrc/gru_seq.hip) -- per-step launches
    there.  One process per GPU (RCCL) keeps the persistent launches."""
    if dp.world_size() > 1 and dp.backend() != "nccl":
        ops.gru_seq_allow(False)


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
        # DTC_GRU_MULTI=1: memory_a and memory_c advance together, one launch per time step (dtc_gru_fwd_multi / dtc_gru_bwd_multi;
        # bit-identical, measured slower: 100.7 vs 92.7 ms per step); default: one chain of launches each, on its lane
        self.gru_multi = os.environ.get("DTC_GRU_MULTI", "0") == "1"
        self._pad_bufs, self._pad_gen = {}, {}
        self._lanes = None
        self._wimages = None
        # every GEMM outside the GRU time steps on block-scaled fp16 operand images (dtc_amd/h2i.py; DTC_H2I=0: round 4's converting
        # kernels): the input projection reads the padded observations' valid rows as an image packed once per update and mini-batch,
        # the MLP activations / gradients live as images, and the weight gradients of a recurrence -- W_ih, W_hh and the MLP layers --
        # are ONE grouped image-operand launch on the weight-gradient stream
        self.use_images = os.environ.get("DTC_H2I", "1") != "0"
        self._imgs = {}                    # name -> h2i.HImage (persistent per name and shape)
        self._wset = None                  # h2i.WeightSet: the step's weight images, rebuilt by one launch per optimisation step
        self._pack_gen, self._pack_slot, self._pack_key = None, 0, {}
        self._wg_ws = {}

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
        adaptive = self.desired_kl is not None and self.schedule == 'adaptive'
        cfg.adaptive_schedule = int(adaptive and dp.world_size() == 1)
        # data parallel: the finalize launch also deposits the KL mean in slot 0 of the gradient header (averaged by the exchange)
        cfg.kl_mirror = self.actor_critic.ensure_arena().kl_slot.data_ptr() if (adaptive and dp.world_size() > 1) else None
        return cfg

    _SEQ_PAIR = os.environ.get("DTC_GRU_SEQ_PAIR", "0") == "1"      # opt-in: measured slower than two lanes

    def _seq_pair(self, T, R, H):
        """The actor's and the critic's forward recurrence go out as ONE persistent launch (dtc_gru_fwd_multi -> dtc_gru_seq_fwd_pair)."""
        return self._SEQ_PAIR and ops.SPLIT and bool(_ffi.lib().dtc_gru_seq_supported(int(T), int(R), int(H), 1))

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
        if self._pack_gen is None:
            # outside update() only.  Inside it the device-side learning rate carries the adaptive schedule from mini-batch to mini-batch
            # (ppo.py:301-307): re-seeding it here from the host copy made every mini-batch adapt from the rate the update STARTED with
            # (found by test_two_consecutive_updates_vs_oracle, round 6)
            self.optimizer.set_lr(self.learning_rate)
        _ffi.lib().dtc_set_concurrency_hint(int(bool(self.overlap)))
        if self._wimages is None:
            self._wimages = ops.WeightImages()
        if self._image_mode(M):
            if self._wset is None:
                self._wset = h2i.WeightSet()
            self._wset.rebuild()                     # the optimiser wrote the weights since the images were built: one grouped launch
            self._forward_backward_images(ln, batch, stats, unpad_idx, store_idx, M, T, R, dev)
        else:
            with self._wimages:                      # weight images of the step's split-path layers: one launch
                self._forward_backward(ln, batch, stats, unpad_idx, store_idx, M, T, R, dev)
        arena = ac.arena
        dp_adaptive = dp.world_size() > 1 and self.desired_kl is not None and self.schedule == 'adaptive'
        if dp.world_size() > 1:
            dp.allreduce_mean_(arena.grad_full)      # header (KL) + every gradient: one collective per optimiser step
        if dp_adaptive:
            ops.lr_adapt(arena.kl_slot, self.optimizer.lr_dev, float(self.desired_kl), kl_out=stats[S_KL:S_KL + 1])
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
        # (dp_adaptive: the KL mean travels in the header of the gradient exchange -- deposited by the loss's finalize launch, _loss_cfg)
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

    # ---------------------------------------------------------------- the same step on operand images
    def _image_mode(self, M):
        ac = self.actor_critic
        return (self.use_images and ops.SPLIT and M % 128 == 0 and ac.memory_a.kind == 'gru' and ac.memory_c.kind == 'gru'
                and ac.memory_a.num_layers == 1 and ac.memory_c.num_layers == 1 and ac.rnn_hidden_size % 128 == 0)

    def _img(self, name, M, K, dev):
        key = (name, int(M), int(K))
        im = self._imgs.get(key)
        if im is None:
            im = self._imgs[key] = h2i.HImage(M, K, dev)
        return im

    def _packed_obs(self, name, x, unpad_idx, M, dev):
        """Image of the valid rows of the padded observations x [T, R, I].  Inside an update the generator yields the SAME
        trajectories for mini-batch i in every epoch (rollout_storage.py:217-267: no shuffling), so each mini-batch's image is packed
        once per update into its own buffer; outside an update (pack_gen None) every call packs."""
        slot = self._pack_slot if self._pack_gen is not None else 0
        im = self._img(f"{name}@{slot}", M, x.shape[-1], dev)
        key = None if self._pack_gen is None else self._pack_gen
        if key is None or self._pack_key.get((name, slot)) != key:
            x2 = x.float().contiguous().view(-1, x.shape[-1])     # (the generator's slice of the padded trajectories: copied only here)
            im.pack(segmat([seg(x2, 0, x2.shape[1], gather=True)], unpad_idx), M)
            self._pack_key[(name, slot)] = key
        return im

    def _forward_backward_images(self, ln, batch, stats, unpad_idx, store_idx, M, T, R, dev):
        ac, st = self.actor_critic, self.storage
        arena, wset = ac.arena, self._wset
        (obs_b, cobs_b, _a, _v, _adv, _r, _lp, _mu, _sg, (hid_a, hid_c), masks) = batch
        H = ac.rnn_hidden_size
        ln.begin(self.overlap)

        # A head runs in three parts -- input projection | recurrence | MLP -- each on the head's lane; with DTC_GRU_MULTI=1 the two
        # recurrences advance TOGETHER on the main lane instead (ops.gru_fwd_multi / gru_bwd_multi: one launch per time step for both).
        multi = self.gru_multi
        # forward only: both recurrences as ONE persistent launch (csrc/gru_seq.hip) where that serves the shape -- opt-in (DTC_GRU_SEQ=1 DTC_GRU_SEQ_PAIR=1); default: two lanes
        multi_fwd = multi or self._seq_pair(T, R, H)

        def head_project(name, mem, layers, x, hidden):
            ximg = self._packed_obs("x_" + name, x, unpad_idx, M, dev)
            gi_c = torch.empty(M, 3 * H, device=dev)
            h2i.linear_fwd(ximg, mem.W_ih, mem.b_ih, gi_c, None, None, wset=wset)
            gi = mem._padded_gi(T * R, 3 * H, dev)
            ops.scatter_rows(gi_c, unpad_idx, gi)
            h0, _ = mem._split(hidden)
            hs_all, gates, hn = torch.empty(T + 1, R, H, device=dev), torch.empty(T, R, 3 * H, device=dev), torch.empty(T, R, H, device=dev)
            ws = ops.workspace(ops.gru_workspace_bytes(T, R, H), dev)
            hd = dict(name=name, mem=mem, layers=layers, ximg=ximg, hs_all=hs_all, gates=gates, hn=hn, ws=ws, gi=gi, h0=h0[0].contiguous(),
                      keep=[gi_c])
            if not multi_fwd:
                ops.gru_fwd(*fwd_item(hd))
            return hd

        def fwd_item(hd):
            return (hd["gi"].view(T, R, 3 * H), hd["h0"], hd["mem"].W_hh, hd["mem"].b_hh, hd["hs_all"], hd["gates"], hd["hn"], hd["ws"])

        def head_mlp(hd):
            name, layers, hs_all = hd["name"], hd["layers"], hd["hs_all"]
            # the MLP reads the un-padded outputs as an image; its hidden activations leave as fp32 (ELU derivative) AND as images
            hx = self._img("hx_" + name, M, H, dev).pack(segmat([seg(hs_all[1:].reshape(T * R, H), 0, H, gather=True)], unpad_idx), M)
            outs, imgs = [], [hx]
            for li, L in enumerate(layers):
                o = torch.empty(M, L.n_out, device=dev)
                oi = self._img(f"o{li}_{name}", M, L.n_out, dev) if li < len(layers) - 1 else None
                h2i.linear_fwd(imgs[-1], L.W, L.b, o, oi, L.act, wset=wset)
                outs.append(o)
                imgs.append(oi)
            hd.update(outs=outs, imgs=imgs)

        def head_mlp_backward(hd, dOut):
            name, mem, layers, outs, imgs = hd["name"], hd["mem"], hd["layers"], hd["outs"], hd["imgs"]
            jobs = hd["jobs"] = []
            dZi = self._img("dout_" + name, M, dOut.shape[1], dev).pack(dOut)
            d_in = torch.empty(M, H, device=dev)
            for li in range(len(layers) - 1, -1, -1):
                L = layers[li]
                jobs.append((dZi, imgs[li], L.gW, 0, L.gb))
                if li > 0:
                    dXi = self._img(f"d{li}_{name}", M, L.n_in, dev)
                    h2i.linear_dgrad(dZi, L.W, None, dXi, Xsaved=outs[li - 1], act=layers[li - 1].act, wset=wset)
                    dZi = dXi
                else:
                    h2i.linear_dgrad(dZi, L.W, d_in, None, wset=wset)
            # one padded buffer per mini-batch slot, zeroed once per update: the slot's trajectories (and padding rows) are the same in
            # every epoch, every scatter overwrites all valid rows
            slot = self._pack_slot if self._pack_gen is not None else 0
            pk = (f"dhs_{name}", slot)
            dhs = self._pad_bufs.get(pk)
            if (dhs is None or tuple(dhs.shape) != (T * R, H) or self._pack_gen is None or self._pad_gen.get(pk) != self._pack_gen):
                if dhs is None or tuple(dhs.shape) != (T * R, H):
                    dhs = self._pad_bufs[pk] = torch.zeros(T * R, H, device=dev)
                else:
                    dhs.zero_()
                self._pad_gen[pk] = self._pack_gen
            ops.scatter_rows(d_in, unpad_idx, dhs)
            dgi, dh0 = torch.empty(T, R, 3 * H, device=dev), torch.empty(R, H, device=dev)
            hd.update(dhs=dhs, dgi=dgi, dh0=dh0)
            hd["keep"] += [d_in, dhs, dgi, dh0, jobs]
            if not multi:
                ops.gru_bwd(dhs.view(T, R, H), hd["hs_all"], hd["gates"], hd["hn"], mem.W_hh, dgi, None, None, dh0, hd["ws"])

        def bwd_item(hd):
            return (hd["dhs"].view(T, R, H), hd["hs_all"], hd["gates"], hd["hn"], hd["mem"].W_hh, hd["dgi"], hd["dh0"], hd["ws"])

        def head_recurrence_grads(hd):
            name, mem, jobs, dgi = hd["name"], hd["mem"], hd["jobs"], hd["dgi"]
            # the recurrence's two weight gradients over the VALID (t, r) slots: dgh / dgi / h_{t-1} rows gathered into images
            # (dgh and dgi share their r / z gate blocks and differ in the n block -- gru_gate_bwd_kernel: da_n vs da_n * r --: the shared
            # 2H columns are packed once, each product runs as two jobs over the row ranges [0, 2H) and [2H, 3H) of its gradient)
            rows = lambda t, c0, w: segmat([seg(t, c0, w, gather=True)], unpad_idx)
            dgh, dgi2 = ops.gru_dgh_all(hd["ws"], T, R, H), dgi.view(T * R, 3 * H)
            rzi = self._img("drz_" + name, M, 2 * H, dev).pack(rows(dgh, 0, 2 * H), M)
            nhi = self._img("dnh_" + name, M, H, dev).pack(rows(dgh, 2 * H, H), M)
            nii = self._img("dni_" + name, M, H, dev).pack(rows(dgi2, 2 * H, H), M)
            hpi = self._img("hp_" + name, M, H, dev).pack(rows(hd["hs_all"][:T].reshape(T * R, H), 0, H), M)
            for gW, gb, X, ni in ((mem.gW_hh, mem.gb_hh, hpi, nhi), (mem.gW_ih, mem.gb_ih, hd["ximg"], nii)):
                jobs.append((rzi, X, gW[:2 * H], 0, gb[:2 * H]))
                jobs.append((ni, X, gW[2 * H:], 0, gb[2 * H:]))
            # one grouped launch on the weight-gradient stream, behind everything this lane has issued
            need = h2i.wgrad_group_workspace_bytes(jobs, M)
            wg = self._wg_ws.get(name)
            if wg is None or wg.numel() * wg.element_size() < need:
                torch.cuda.synchronize()
                wg = self._wg_ws[name] = ops.workspace(need, dev)
            if self.overlap:
                ev = ln.event()
                ev.record()
                ln.side.wait_event(ev)
                h2i.wgrad_group(jobs, M, wg, stream_ptr=ln.side.cuda_stream)
                ln.side_busy = True
            else:
                h2i.wgrad_group(jobs, M, wg)

        with ln.lane("aux"):
            hc = head_project("c", ac.memory_c, ac.Cr, cobs_b, hid_c)
        ha = head_project("a", ac.memory_a, ac.A, obs_b, hid_a)
        if multi_fwd:
            ln.order("aux", "main")
            ops.gru_fwd_multi([fwd_item(ha), fwd_item(hc)])
            ln.order("main", "aux")
        with ln.lane("aux"):
            head_mlp(hc)
        head_mlp(ha)
        ln.order("aux", "main")
        mean, value = ha["outs"][-1], hc["outs"][-1]
        ac._dist = (mean, ac.std_view.detach().expand_as(mean))
        ac._actor_outs, ac._critic_outs = ha["outs"], hc["outs"]
        ac.memory_a.saved = dict(hs_all=ha["hs_all"], out=ha["hs_all"][1:])
        ac.memory_c.saved = dict(hs_all=hc["hs_all"], out=hc["hs_all"][1:])
        dmean, dval = torch.empty_like(mean), torch.empty(M, 1, device=dev)
        lws = ops.workspace(_ffi.lib().dtc_loss_workspace(M), dev)
        flat = lambda k: st.flat(k)
        ops.ppo_loss(mean, ac.std_view, value, flat("actions"), flat("actions_log_prob"), flat("mu"), flat("sigma"),
                     flat("advantages"), flat("returns"), flat("values"), store_idx, self._loss_cfg(), dmean, dval,
                     ac.std_grad, stats[S_SURR:S_SURR + 4], self.optimizer.lr_dev, lws)
        # (data parallel: the KL mean travels in the header of the gradient exchange -- deposited by the loss's finalize launch, _loss_cfg)
        ln.order("main", "aux")
        with ln.lane("aux"):
            head_mlp_backward(hc, dval)
        head_mlp_backward(ha, dmean)
        if multi:
            ln.order("aux", "main")
            ops.gru_bwd_multi([bwd_item(ha), bwd_item(hc)])
            ln.order("main", "aux")
        with ln.lane("aux"):
            head_recurrence_grads(hc)
        head_recurrence_grads(ha)
        ln.join()
        self._held = (ha, hc, dmean, dval, lws)                 # (until the next step: nothing here returns to the allocator early)

    def update(self):
        self._require_gpu()
        _share_rule()
        st = self.storage
        nmb, epochs = self.num_mini_batches, self.num_learning_epochs
        mb = st.num_envs // nmb
        dev = self.actor_critic.std.device
        stats = torch.zeros(nmb * epochs, STAT_COLS, device=dev)
        k = 0
        for mem in (self.actor_critic.memory_a, self.actor_critic.memory_c):
            mem.new_update()
        # the observation images of this update: packed once per mini-batch.  The generation number must never repeat: `_pack_gen` is
        # None between updates, so it is drawn from a counter of its own (it used to be `(self._pack_gen or 0) + 1` = 1 in EVERY
        # update: from the second update of a run on, the image path reused the first update's packed observations)
        self._pack_serial = getattr(self, "_pack_serial", 0) + 1
        self._pack_gen = self._pack_serial
        self.optimizer.set_lr(self.learning_rate)      # once per update: the schedule then lives on the device (lr_dev)
        try:
            for batch in st.reccurent_mini_batch_generator(nmb, epochs):
                i = k % nmb
                self._pack_slot = i
                self.step_minibatch(batch, i * mb, (i + 1) * mb, stats[k])
                k += 1
        finally:
            self._pack_gen = None
        host = stats.cpu()
        ops.gru_seq_check()                      # (the persistent recurrence launches of this update all ran to their end)
        self.learning_rate = float(self.optimizer.lr_dev.item())
        self.last_update_stats = host
        m = host.double().mean(dim=0)
        st.clear()
        return float(m[S_VALUE]), float(m[S_SURR])
