"""PPO for the GRU + CE-net composite (ActorCriticDecoderRecurrent) -- BASELINE.json config 5.

Per recurrent mini-batch (N/4 envs x all 24 steps, rollout_storage.py:217-267) exactly the two optimisation steps
of ppo.py:189-338 (SURVEY.md §8a "Config 5"):
  1. the VAE step of `PPO` (inherited unchanged) on the valid (t, env) rows of the env slice -- the outlier
     statistics of the CE-net never see padding;
  2. the policy step with BPTT: CE-net / terrain encoder features -> GRU input projections on the valid rows
     (feature blocks are segments of the GEMM operand, nothing is concatenated) -> row scatter into the padded
     [T, n_traj] layout -> dtc_gru_fwd -> MLP heads with the un-padding folded in as a row gather -> fused PPO loss
     -> MLP backward -> row scatter -> dtc_gru_bwd -> row gather -> input-projection backward, whose data gradient
     fans out to z, mu[:, :3] and l_t -> CE-net / terrain encoder backward -> clip + Adam.
Hidden states are recorded BEFORE each rollout step (the convention of the commented lines ppo.py:138-139) and the
mini-batch takes the states at its trajectory starts, as `reccurent_mini_batch_generator` does.
Same constructor keywords / method names as `PPO`; weight gradients run on the side stream (see PPO._bwd).
"""
from __future__ import annotations

import os

import torch

from .. import _ffi, distributed as dp, h2i, ops
from .._ffi import seg, segmat
from ..modules.actor_critic_decoder import Dense
from ..modules.actor_critic_decoder_recurrent import ActorCriticDecoderRecurrent
from ..utils import split_and_pad_trajectories, true_indices
from .ppo import PPO, S_GNORM, S_KL, S_RECONS, S_SURR, S_VALUE, S_VEL, S_KLD, STAT_COLS
from .recurrent_ppo import _share_rule


class RecurrentDecoderPPO(PPO):
    actor_critic: ActorCriticDecoderRecurrent
    # DTC_GRU_MULTI=1: the actor's and the critic's recurrence advance together, one launch per time step (dtc_gru_fwd_multi /
    # dtc_gru_bwd_multi; bit-identical).  Off: measured 136 vs 129 ms per step against one chain of launches per recurrence, each on
    # its own lane
    gru_multi = os.environ.get("DTC_GRU_MULTI", "0") == "1"

    # ---------------------------------------------------------------- rollout side
    def act(self, obs, privileged_obs, obs_history, base_vel, rew_buf=None):
        self._require_gpu()
        ac = self.actor_critic
        ac.ensure_arena()
        ac._ensure_hidden(obs.shape[0], obs.device)
        hidden = tuple(h.clone() for h in ac.get_hidden_states())          # state BEFORE this step
        actions = super().act(obs, privileged_obs, obs_history, base_vel, rew_buf)
        self.transition.hidden_states = hidden
        return actions

    def compute_returns(self, last_critic_obs, last_critic_privileged_obs, last_base_vel):
        self._require_gpu()
        ac = self.actor_critic
        keep = ac.memory_c.hidden_states.clone() if ac.memory_c.hidden_states is not None else None
        last_values = ac.evaluate(last_critic_obs, last_critic_privileged_obs, last_base_vel).detach()
        ac.memory_c.hidden_states = keep            # the bootstrap value must not advance the critic's state
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    # ---------------------------------------------------------------- recurrent mini-batches as index data
    def recurrent_slices(self, hid_a=None, hid_c=None):
        """Yield, per mini-batch of envs [a, b): time-major flat row indices of its samples, the padded-layout
        row of each sample (`unpad_idx`), T, n_traj and the hidden states at the trajectory starts."""
        st = self.storage
        T, N = st.num_transitions_per_env, st.num_envs
        dev = st.dones.device
        hid_a = st.saved_hidden_states_a[0] if hid_a is None else hid_a
        hid_c = st.saved_hidden_states_c[0] if hid_c is None else hid_c
        mb = N // self.num_mini_batches
        _, masks_all = split_and_pad_trajectories(st.dones, st.dones)
        dones = st.dones.squeeze(-1)
        lwd = torch.zeros_like(dones, dtype=torch.bool)
        lwd[1:] = dones[:-1].bool()
        lwd[0] = True
        counts = lwd.view(T, self.num_mini_batches, mb).sum(dim=(0, 2)).tolist()     # one host sync per update
        first = 0
        for i in range(self.num_mini_batches):
            a, b = i * mb, (i + 1) * mb
            last = first + int(counts[i])
            masks = masks_all[:, first:last]
            R = last - first
            flat_rt = true_indices(masks.transpose(1, 0), mb * T)      # no nonzero() sync: the count is known
            traj, pos = flat_rt // T, flat_rt % T
            unpad_idx = (pos * R + traj).view(mb, T).transpose(1, 0).reshape(-1).contiguous()
            idx = (torch.arange(T, device=dev).unsqueeze(1) * N + torch.arange(a, b, device=dev)).reshape(-1).contiguous()
            starts = true_indices(lwd[:, a:b].permute(1, 0), R)      # (env, t) of every trajectory start, env-major; count known
            s_env, s_t = a + starts // T, starts % T
            pick = lambda h: h[s_t, 0, s_env].contiguous()           # [R, H]: layer 0's saved state at every trajectory start
            yield dict(a=a, b=b, idx=idx, unpad_idx=unpad_idx, T=T, R=R, hid_a=pick(hid_a), hid_c=pick(hid_c))
            first = last

    # ---------------------------------------------------------------- policy step with BPTT
    def _padded(self, tw, name, rows, width):
        key = ("pad", name)
        t = tw._g.get(key)
        if t is None or t.shape[0] != rows:
            t = tw._g[key] = torch.zeros(rows, width, dtype=torch.float32, device=tw._dev)
        return t

    def _ppo_step_recurrent(self, fw, tw, flat, bt, eps, stats, cfg):
        if self._image_mode(fw):
            # every GEMM outside the GRU time steps on operand images (dtc_amd/h2i.py), as in PPO._ppo_step
            tw.narrow_wgrad = True
            wset = self._wset("ppo_recurrent")
            wset.rebuild()
            early = self._ppo_recurrent_forward_backward_images(fw, tw, flat, bt, eps, stats, cfg, wset)
        else:
            with self._images("ppo_recurrent"):                    # weight images of the step's split-path layers: one launch
                early = self._ppo_recurrent_forward_backward(fw, tw, flat, bt, eps, stats, cfg)
        if not early:
            self._allreduce_grads(self.optimizer)
        self._lr_from_header(stats)
        if self.capture_grads:
            self.captured["main"] = self.actor_critic.arena.grad.clone()
        self.optimizer.step(self.max_grad_norm, stats[S_GNORM:S_GNORM + 1])

    def _ppo_recurrent_forward_backward(self, fw, tw, flat, bt, eps, stats, cfg):
        ac = self.actor_critic
        H, M = ac.rnn_hidden_size, tw.B
        idx, unpad_idx, T, R = bt["idx"], bt["unpad_idx"], bt["T"], bt["R"]
        dev = idx.device
        # lanes: the critic head (raw-input features -> GRU -> MLP) is independent of the CE-net / terrain encoders and
        # of the actor head until the loss, and again until the optimiser step: it runs on `aux`, the small per-time-
        # step kernels of the two recurrences overlap
        _ffi.lib().dtc_set_concurrency_hint(int(bool(self.overlap_wgrad)))
        tw.begin(self.overlap_lanes and self.overlap_wgrad)
        Xa = ac.actor_input(fw, flat["observations"], idx)
        Xc = ac.critic_input(flat["observations"], flat["base_vel"], flat["privileged_observations"], idx)

        def head_forward(name, X, mem, proj, layers, h0):
            gi_v = tw.g("gi_" + name, 3 * H)
            ops.linear_fwd(X, proj.W, proj.b, gi_v, None, M=M)
            gi_p = self._padded(tw, "gi_" + name, T * R, 3 * H)     # padded steps keep finite stale values (never used)
            ops.scatter_rows(gi_v, unpad_idx, gi_p)
            hs_all = torch.empty(T + 1, R, H, device=dev)
            gates, hn = torch.empty(T, R, 3 * H, device=dev), torch.empty(T, R, H, device=dev)
            ws = ops.workspace(ops.gru_workspace_bytes(T, R, H), dev)
            ops.gru_fwd(gi_p.view(T, R, 3 * H), h0.contiguous(), mem.W_hh, mem.b_hh, hs_all, gates, hn, ws)
            X0 = segmat([seg(hs_all[1:].reshape(T * R, H), 0, H, gather=True)], unpad_idx)
            outs, cur = [], X0
            for li, L in enumerate(layers):
                o = tw.g(f"{name}_o{li}", L.n_out)
                ops.linear_fwd(cur, L.W, L.b, o, L.act, M=M)
                outs.append(o)
                cur = o
            return dict(name=name, X=X, mem=mem, proj=proj, layers=layers, hs_all=hs_all, gates=gates, hn=hn, ws=ws,
                        X0=X0, outs=outs)

        def head_backward(hd, dOut):
            name, layers, outs = hd["name"], hd["layers"], hd["outs"]
            dZ = dOut
            for li in range(len(layers) - 1, -1, -1):
                dX = tw.g(f"{name}_d{li}", layers[li].n_in)
                if li > 0:
                    self._bwd(tw, layers[li], dZ, outs[li - 1], dX, outs[li - 1], layers[li - 1].act)
                else:
                    self._bwd(tw, layers[li], dZ, hd["X0"], dX, None, None)
                dZ = dX
            dhs = self._padded(tw, "dhs_" + name, T * R, H)
            dhs.zero_()
            ops.scatter_rows(dZ, unpad_idx, dhs)
            dgi_p, dh0 = torch.empty(T, R, 3 * H, device=dev), torch.empty(R, H, device=dev)
            mem = hd["mem"]
            ops.gru_bwd(dhs.view(T, R, H), hd["hs_all"], hd["gates"], hd["hn"], mem.W_hh, dgi_p, mem.gW_hh, mem.gb_hh,
                        dh0, hd["ws"], rows=unpad_idx)          # the W_hh weight gradient skips the padding slots
            return ops.gather_rows(dgi_p.view(T * R, 3 * H), unpad_idx, out=tw.g("dgi_" + name, 3 * H))

        with tw.lane("aux"):
            hc = head_forward("c", Xc, ac.memory_c, ac.proj_c, ac.Cr, bt["hid_c"])
        ac.cenet_forward_(fw, flat["observation_histories"], eps, idx, masks=self.relu_masks)
        ac.terrain_encoder_(fw, flat["privileged_observations"], idx, masks=self.relu_masks)
        ha = head_forward("a", Xa, ac.memory_a, ac.proj_a, ac.A, bt["hid_a"])
        tw.order("aux", "main")
        if self.after_forward_hook is not None:
            self.after_forward_hook(fw, "ppo")
        mean, value = ha["outs"][-1], hc["outs"][-1]
        ops.ppo_loss(mean, ac.std_view, value, flat["actions"], flat["actions_log_prob"], flat["mu"], flat["sigma"],
                     flat["advantages"], flat["returns"], flat["values"], idx, cfg, tw.dmean, tw.dval, ac.std_grad,
                     stats[S_SURR:S_SURR + 4], self.optimizer.lr_dev, tw.loss_ws)
        self._kl_to_header(stats)
        tw.order("main", "aux")
        with tw.lane("aux"):
            self._bwd(tw, hc["proj"], head_backward(hc, tw.dval), hc["X"])
        dgi_a = head_backward(ha, tw.dmean)
        # the actor features' gradient fans out to z, mu[:, :3], l_t (observations need none)
        tw.dmulv.zero_()
        dst = segmat([seg(None, 0, ac.num_obs), seg(tw.dz, 0, 16), seg(tw.dmulv, 0, 3), seg(tw.dlt, 0, 512)])
        self._bwd(tw, ha["proj"], dgi_a, ha["X"], dst, None, None)
        # every gradient of the first bucket (both MLP heads, both GRUs, both input projections, std) has been written or
        # queued: it travels (with the KL header) while the encoders run backward
        early = self._exchange_bucket(tw, "main_only")
        ops.cenet_latent_bwd(tw.dmulv, tw.dz, eps, fw.mulv, fw.mask, fw.info, fw.lat_ws)
        self._terrain_encoder_backward(fw, tw, flat, idx)
        self._cenet_encoder_backward(fw, tw, flat, idx)
        if early:
            self._exchange_bucket(tw, "shared")
        self._join(tw)
        return early

    def _ppo_recurrent_forward_backward_images(self, fw, tw, flat, bt, eps, stats, cfg, wset):
        """The policy step with BPTT on operand images: encoders as in PPO._ppo_forward_backward, the GRU input projections read their
        feature blocks as images (the critic's packed once per update and mini-batch; the actor's = the l_t image + the packed
        observations + the latent kernel's [z | mu[:, :3]] image), the MLP heads live on images, and the weight gradients -- MLPs, W_hh
        (dgh_all gathered from dtc_gru_bwd's workspace), W_ih by feature block -- join the bucket's grouped image launches."""
        ac = self.actor_critic
        H, M = ac.rnn_hidden_size, tw.B
        idx, unpad_idx, T, R = bt["idx"], bt["unpad_idx"], bt["T"], bt["R"]
        dev = idx.device
        obs, priv = flat["observations"], flat["privileged_observations"]
        imn = self.narrow_images
        ns = False
        _ffi.lib().dtc_set_concurrency_hint(int(bool(self.overlap_wgrad)))
        tw.begin(self.overlap_lanes and self.overlap_wgrad)
        tw.live_img.clear()
        rows = lambda t, w: segmat([seg(t, 0, w, gather=True)], unpad_idx)
        # forward only: both recurrences as ONE persistent launch (csrc/gru_seq.hip) where that serves the shape -- opt-in (DTC_GRU_SEQ=1 DTC_GRU_SEQ_PAIR=1); default: two lanes
        multi_fwd = self.gru_multi or (os.environ.get("DTC_GRU_SEQ_PAIR", "0") == "1" and ops.SPLIT and
                                       bool(_ffi.lib().dtc_gru_seq_supported(int(T), int(R), int(H), 1)))

        # A head runs in three parts -- input projection | recurrence | MLP -- each on the head's lane; with DTC_GRU_MULTI=1 the two
        # recurrences advance TOGETHER on the main lane instead (ops.gru_fwd_multi / gru_bwd_multi: one launch per time step for both).
        def head_project(name, X, cols, mem, proj, layers, h0):
            gi_v = tw.g("gi_" + name, 3 * H)
            h2i.linear_fwd(X, proj.W, proj.b, gi_v, None, None, wset=wset, cols=cols)
            gi_p = self._padded(tw, "gi_" + name, T * R, 3 * H)     # padded steps keep finite stale values (never used)
            ops.scatter_rows(gi_v, unpad_idx, gi_p)
            hs_all = torch.empty(T + 1, R, H, device=dev)
            gates, hn = torch.empty(T, R, 3 * H, device=dev), torch.empty(T, R, H, device=dev)
            ws = ops.workspace(ops.gru_workspace_bytes(T, R, H), dev)
            hd = dict(name=name, X=X, cols=cols, mem=mem, proj=proj, layers=layers, hs_all=hs_all, gates=gates, hn=hn, ws=ws,
                      gi_p=gi_p, h0=h0.contiguous())
            if not multi_fwd:
                ops.gru_fwd(*fwd_item(hd))
            return hd

        def fwd_item(hd):
            return (hd["gi_p"].view(T, R, 3 * H), hd["h0"], hd["mem"].W_hh, hd["mem"].b_hh, hd["hs_all"], hd["gates"], hd["hn"], hd["ws"])

        def head_mlp(hd):
            name, layers, hs_all = hd["name"], hd["layers"], hd["hs_all"]
            hx = tw.img("hx_" + name, H).pack(rows(hs_all[1:].reshape(T * R, H), H), M)
            outs, imgs = [], [hx]
            for li, L in enumerate(layers):
                o = tw.g(f"{name}_o{li}", L.n_out)
                oi = tw.img(f"{name}_o{li}", L.n_out) if li < len(layers) - 1 else None
                h2i.linear_fwd(imgs[-1], L.W, L.b, o, oi, L.act, wset=wset)
                outs.append(o)
                imgs.append(oi)
            hd.update(outs=outs, imgs=imgs)
            return hd

        def head_mlp_backward(hd, dOut):
            """MLP backward down to the padded gradient of the recurrence's outputs."""
            name, layers, outs, imgs, mem = hd["name"], hd["layers"], hd["outs"], hd["imgs"], hd["mem"]
            dZi = tw.img("dout_" + name, dOut.shape[1]).pack(dOut)
            d_in = tw.g(f"{name}_d0", H)
            for li in range(len(layers) - 1, -1, -1):
                L = layers[li]
                self._bwd_img(tw, L, dZi, imgs[li])
                if li > 0:
                    dXi = tw.img(f"{name}_d{li}", L.n_in)
                    h2i.linear_dgrad(dZi, L.W, None, dXi, Xsaved=outs[li - 1], act=layers[li - 1].act, wset=wset)
                    dZi = dXi
                else:
                    h2i.linear_dgrad(dZi, L.W, d_in, None, wset=wset)
            # one padded buffer per mini-batch slot: the slot's trajectories -- and with them its padding rows -- are the same in every
            # epoch of an update (fw.pack_gen), the valid rows are overwritten by every scatter, so the padding is zeroed once per
            # update and slot instead of once per mini-batch (32 of 40 fills of 72 MB per step)
            pkey = f"dhs_{name}_{fw.pack_slot}"
            dhs = self._padded(tw, pkey, T * R, H)
            gens = tw.__dict__.setdefault("_pad_gen", {})
            if fw.pack_gen is None or gens.get(pkey) != (fw.pack_gen, T * R):
                dhs.zero_()
                gens[pkey] = (fw.pack_gen, T * R)
            ops.scatter_rows(d_in, unpad_idx, dhs)
            dgi_p, dh0 = torch.empty(T, R, 3 * H, device=dev), torch.empty(R, H, device=dev)
            hd.update(dhs=dhs, dgi_p=dgi_p, dh0=dh0)
            tw.held.append((dgi_p, dh0, hd))
            if not self.gru_multi:
                ops.gru_bwd(dhs.view(T, R, H), hd["hs_all"], hd["gates"], hd["hn"], mem.W_hh, dgi_p, None, None, dh0, hd["ws"])

        def bwd_item(hd):
            return (hd["dhs"].view(T, R, H), hd["hs_all"], hd["gates"], hd["hn"], hd["mem"].W_hh, hd["dgi_p"], hd["dh0"], hd["ws"])

        def head_recurrence_grads(hd, full_dgi=True):
            """Behind the BPTT: the W_hh weight gradient (queued); returns the image(s) of dgi over the valid rows: the whole [M, 3H]
            image (the actor: its input projection's data gradient reduces over all 3H columns), or (full_dgi=False, the critic) the
            pair (r / z blocks [M, 2H], n block [M, H]) -- dgh and dgi share their r / z blocks (gru_gate_bwd_kernel: da_n vs da_n * r
            in the n block only), so those 2H columns are packed once and each weight gradient runs as two jobs over row ranges."""
            name, mem, dgi_p = hd["name"], hd["mem"], hd["dgi_p"]
            hpi = tw.img("hp_" + name, H).pack(rows(hd["hs_all"][:T].reshape(T * R, H), H), M)
            dgh = ops.gru_dgh_all(hd["ws"], T, R, H)
            if full_dgi:
                dghi = tw.img("dgh_" + name, 3 * H).pack(rows(dgh, 3 * H), M)
                dgii = tw.img("dgi_" + name, 3 * H).pack(rows(dgi_p.view(T * R, 3 * H), 3 * H), M)
                self._bwd_img(tw, Dense(mem.W_hh, mem.b_hh, mem.gW_hh, mem.gb_hh, None), dghi, hpi)
                return dgii
            cols = lambda t, c0, w: segmat([seg(t, c0, w, gather=True)], unpad_idx)
            rzi = tw.img("drz_" + name, 2 * H).pack(cols(dgh, 0, 2 * H), M)
            nhi = tw.img("dnh_" + name, H).pack(cols(dgh, 2 * H, H), M)
            nii = tw.img("dni_" + name, H).pack(cols(dgi_p.view(T * R, 3 * H), 2 * H, H), M)
            part = lambda W, b, gW, gb, lo, hi: Dense(W[lo:hi], b[lo:hi], gW[lo:hi], gb[lo:hi], None)
            self._bwd_img(tw, part(mem.W_hh, mem.b_hh, mem.gW_hh, mem.gb_hh, 0, 2 * H), rzi, hpi)
            self._bwd_img(tw, part(mem.W_hh, mem.b_hh, mem.gW_hh, mem.gb_hh, 2 * H, 3 * H), nhi, hpi)
            return rzi, nii

        with tw.lane("aux"):
            Xc = ac.packed_input(fw, "p_c", ac.critic_input(obs, flat["base_vel"], priv, idx), idx, reuse=True)
            hc = head_project("c", Xc, None, ac.memory_c, ac.proj_c, ac.Cr, bt["hid_c"])
        ac.cenet_forward_(fw, flat["observation_histories"], eps, idx, masks=self.relu_masks, split=ns, images=imn, wset=wset)
        ac.terrain_encoder_(fw, priv, idx, masks=self.relu_masks, images=True, wset=wset, lt_fp32=False)
        if imn:        # the actor's features as three images: l_t, the gathered observations (packed once), the latent kernel's [z | mu[:, :3]]
            Xa = [fw.img("lt"), ac.packed_input(fw, "p_obs", segmat([seg(obs, 0, ac.num_obs, gather=True)], idx), idx, reuse=True), fw.cur["p_zmu"]]
            a_cols = [ac.num_obs + 19, 0, ac.num_obs]
        else:
            Xa = [fw.img("lt"), ac.packed_input(fw, "p_a", segmat([seg(obs, 0, ac.num_obs, gather=True), seg(fw.z, 0, 16), seg(fw.mulv, 0, 3)], idx))]
            a_cols = [ac.num_obs + 19, 0]
        ha = head_project("a", Xa, a_cols, ac.memory_a, ac.proj_a, ac.A, bt["hid_a"])
        if multi_fwd:
            tw.order("aux", "main")                                 # the critic's input projection is written
            ops.gru_fwd_multi([fwd_item(ha), fwd_item(hc)])
            tw.order("main", "aux")
        with tw.lane("aux"):
            head_mlp(hc)
        head_mlp(ha)
        tw.order("aux", "main")
        if self.after_forward_hook is not None:
            self.after_forward_hook(fw, "ppo")
        mean, value = ha["outs"][-1], hc["outs"][-1]
        ops.ppo_loss(mean, ac.std_view, value, flat["actions"], flat["actions_log_prob"], flat["mu"], flat["sigma"],
                     flat["advantages"], flat["returns"], flat["values"], idx, cfg, tw.dmean, tw.dval, ac.std_grad,
                     stats[S_SURR:S_SURR + 4], self.optimizer.lr_dev, tw.loss_ws)
        self._kl_to_header(stats)
        tw.order("main", "aux")
        with tw.lane("aux"):
            head_mlp_backward(hc, tw.dval)
        head_mlp_backward(ha, tw.dmean)
        if self.gru_multi:
            tw.order("aux", "main")
            ops.gru_bwd_multi([bwd_item(ha), bwd_item(hc)])
            tw.order("main", "aux")
        with tw.lane("aux"):
            rzi_c, nii_c = head_recurrence_grads(hc, full_dgi=False)
            pc = hc["proj"]
            self._bwd_img(tw, Dense(pc.W[:2 * H], pc.b[:2 * H], pc.gW[:2 * H], pc.gb[:2 * H], None), rzi_c, Xc)
            self._bwd_img(tw, Dense(pc.W[2 * H:], pc.b[2 * H:], pc.gW[2 * H:], pc.gb[2 * H:], None), nii_c, Xc)
        dgii_a = head_recurrence_grads(ha)
        # the actor features' gradient fans out to z, mu[:, :3] (fp32) and l_t (image); the observations need none
        tw.dmulv.zero_()
        nb = ac.num_obs + 19
        for i, (xi, c0) in enumerate(zip(Xa, a_cols)):
            self._bwd_img(tw, ha["proj"], dgii_a, xi, wcol0=c0, bias=i == 0)
        h2i.linear_dgrad(dgii_a, ha["proj"].W, segmat([seg(None, 0, 512), seg(tw.dz, 0, 16), seg(tw.dmulv, 0, 3)]), tw.img("dlt", 512),
                         window=[(nb, 512), (ac.num_obs, 19)], wset=wset)
        tw.live_img |= {"dlt"}
        early = self._exchange_bucket(tw, "main_only")
        tw.order("main", "aux")                                    # dz, d mu[:, :3] are written
        self._terrain_encoder_backward(fw, tw, flat, idx, wset)
        with tw.lane("aux"):
            ops.cenet_latent_bwd(tw.dmulv, tw.dz, eps, fw.mulv, fw.mask, fw.info, fw.lat_ws)
            self._cenet_encoder_backward(fw, tw, flat, idx, split=ns, wset=wset if imn else None)
        if early:
            self._exchange_bucket(tw, "shared")
        self._join(tw)
        return early

    def step_minibatch(self, bt, eps1, eps2, which="both", stats=None):
        """One recurrent mini-batch `bt` (an item of `recurrent_slices`): VAE step, policy step, or both."""
        self._require_gpu()
        st, ac = self.storage, self.actor_critic
        self._arena()
        dev = ac.std.device
        B = bt["idx"].numel()
        flat = {k: st.flat(k) for k in self._FLAT_NAMES}
        fw, tw = ac._fwd_ws(B), self._train_ws(B)
        own_gen = fw.pack_gen is None                 # outside update(): the packed rollout rows serve this call only
        if own_gen:
            self._amax_static(flat)                   # (inside update(): once per update -- the storage does not change between mini-batches)
            self._pack_gen = getattr(self, "_pack_gen", 0) + 1
            fw.pack_gen, fw.pack_slot = self._pack_gen, 0
            # (inside update() the device-side learning rate carries the adaptive schedule from mini-batch to mini-batch, ppo.py:301-307:
            # re-seeding it here from the host copy made every mini-batch adapt from the rate the update STARTED with -- found by
            # test_two_consecutive_updates_vs_oracle, round 6)
            self.optimizer.set_lr(self.learning_rate)
        stats = torch.zeros(STAT_COLS, dtype=torch.float32, device=dev) if stats is None else stats
        if which in ("vae", "both"):
            self._vae_step(fw, tw, flat, bt["idx"], eps1.to(dev).contiguous(), stats)
        if which in ("ppo", "both"):
            self._ppo_step_recurrent(fw, tw, flat, bt, eps2.to(dev).contiguous(), stats, self._loss_cfg())
        if own_gen:
            ops.amax_static_clear()
            fw.pack_gen = None
        return stats

    def update(self, eps1=None, eps2=None, return_stats=False):
        self._require_gpu()
        _share_rule()
        st, ac = self.storage, self.actor_critic
        self._arena()
        dev = ac.std.device
        nmb, epochs = self.num_mini_batches, self.num_learning_epochs
        B = (st.num_envs // nmb) * st.num_transitions_per_env
        steps = nmb * epochs
        if eps1 is None or eps2 is None:
            seed = ops.draw_seed()
            eps1 = ops.randn((steps, B, 16), dev, seed + 1) if eps1 is None else eps1
            eps2 = ops.randn((steps, B, 16), dev, seed + 2) if eps2 is None else eps2
        stats = torch.zeros(steps, STAT_COLS, dtype=torch.float32, device=dev)
        for key, buf in self._train_ws(B)._g.items():         # padded input projections: the padding slots start every update from zero
            if isinstance(key, tuple) and key[0] == "pad" and key[1].startswith("gi_"):
                buf.zero_()
        slices = list(self.recurrent_slices())
        k = 0
        fw = ac._fwd_ws(B)
        self.optimizer.set_lr(self.learning_rate)      # once per update: the schedule then lives on the device (lr_dev)
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1
        fw.pack_gen = self._pack_gen                   # the slices are the same in every epoch: their packed rollout rows serve all five
        # amax records of the stored rollout tensors ONCE per update (they were recomputed by every mini-batch: 100 passes over up to
        # 546 MB = 6.5 ms of a 133 ms step)
        self._amax_static({k: st.flat(k) for k in self._FLAT_NAMES})
        try:
            for _ in range(epochs):
                for i, bt in enumerate(slices):
                    fw.pack_slot = i
                    self.step_minibatch(bt, eps1[k], eps2[k], "both", stats[k])
                    k += 1
        finally:
            fw.pack_gen = None
            ops.amax_static_clear()              # the storage is about to be refilled: its amax slots are void
        host = stats.cpu()                       # the single device -> host synchronisation of the update
        ops.gru_seq_check()                      # (the persistent recurrence launches of this update all ran to their end)
        self.learning_rate = float(self.optimizer.lr_dev.item())
        for g in self.optimizer.param_groups:
            g['lr'] = self.learning_rate
        self.last_update_stats = host
        m = host.double().mean(dim=0)
        st.clear()
        out = (float(m[S_VALUE]), float(m[S_SURR]), 0.0, 0, float(m[S_RECONS]), float(m[S_VEL]), float(m[S_KLD]))
        return (out, host) if return_stats else out
