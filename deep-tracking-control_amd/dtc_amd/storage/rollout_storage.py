"""RolloutStorage with the reference's surface (rsl_rl/rsl_rl/storage/rollout_storage.py:36-267)
on HIP kernels.

Buffers are the same public `[T, N, d]` time-major tensors (a1 in SURVEY.md §8a).  What changes:
  * `compute_returns`  -> one GAE-scan kernel + two tiny reduction kernels (rollout_storage.py:138-152);
    the two batch statistics are all-reduced when torch.distributed is initialised (data parallel);
  * `mini_batch_generator` -> row-gather kernels; the permutation is drawn on the device once per
    update and reused by all epochs exactly as rollout_storage.py:165 does.  PPO.update does NOT use
    this generator on its fast path: it hands the permutation to the GEMM operand loaders instead
    (the gathered mini-batch is never materialised); the generator exists for API parity.
"""
from __future__ import annotations

import torch

from .. import distributed as dp
from .. import _ffi, ops
from ..utils import split_and_pad_trajectories, trajectory_index_map


class RolloutStorage:
    class Transition:
        def __init__(self):
            self.observations = None
            self.next_observations = None
            self.privileged_observations = None
            self.observation_histories = None
            self.critic_observations = None
            self.actions = None
            self.rewards = None
            self.dones = None
            self.values = None
            self.actions_log_prob = None
            self.action_mean = None
            self.action_sigma = None
            self.hidden_states = None
            self.base_vel = None

        def clear(self):
            self.__init__()

    def __init__(self, num_envs, num_transitions_per_env, obs_shape, privileged_obs_shape, obs_history_shape,
                 actions_shape, device='cpu'):
        self.device = device
        self.obs_shape = obs_shape
        self.privileged_obs_shape = privileged_obs_shape
        self.obs_history_shape = obs_history_shape
        self.actions_shape = actions_shape
        T, N = num_transitions_per_env, num_envs
        z = lambda *s: torch.zeros(T, N, *s, device=self.device)
        # Core
        self.observations = z(*obs_shape)
        self.next_observations = z(*obs_shape)
        self.privileged_observations = z(*privileged_obs_shape)
        self.observation_histories = z(*obs_history_shape)
        self.rewards = z(1)
        self.actions = z(*actions_shape)
        self.dones = torch.zeros(T, N, 1, device=self.device, dtype=torch.uint8)
        # For PPO
        self.actions_log_prob = z(1)
        self.values = z(1)
        self.returns = z(1)
        self.advantages = z(1)
        self.mu = z(*actions_shape)
        self.sigma = z(*actions_shape)
        self.base_vel = z(3)
        self.num_transitions_per_env = T
        self.num_envs = N
        # rnn
        self.saved_hidden_states_a = None
        self.saved_hidden_states_c = None
        self.step = 0
        self._stats = None

    def add_transitions(self, transition: "RolloutStorage.Transition", time_outs=None, gamma=0.0):
        """rollout_storage.py:99-116.  On a HIP device the 13 copies are ONE dtc_store_transition launch, which also
        applies the time-out bootstrap of PPO.process_env_step (ppo.py:162-163) when `time_outs` is given."""
        if self.step >= self.num_transitions_per_env:
            raise AssertionError("Rollout buffer overflow")
        s, tr = self.step, transition
        if self.observations.is_cuda:
            f = lambda t: t if t.dtype == torch.float32 else t.float()
            dones = tr.dones if tr.dones.dtype in (torch.uint8, torch.bool) else tr.dones.to(torch.uint8)
            if dones.dtype == torch.bool:
                dones = dones.view(torch.uint8)
            # the 11 destination rows of step s: addresses from a plan marshalled once (base pointer + s * step bytes),
            # only the source pointers are filled in per step -- the env step is host-bound, every us of FFI glue counts
            plan = self._store_plan()
            srcs = (f(tr.observations), f(tr.next_observations), f(tr.privileged_observations), f(tr.observation_histories),
                    f(tr.actions), dones.reshape(-1), f(tr.values), f(tr.actions_log_prob).reshape(-1), f(tr.action_mean),
                    f(tr.base_vel), f(tr.action_sigma))
            items = plan["items"]
            for i, (src, (base, step_bytes, width, numel, dtype)) in enumerate(zip(srcs, plan["dst"])):
                # per-call checks (a handful of integer compares): the kernel trusts these descriptors blindly
                if src.shape[0] != self.num_envs or src.numel() != self.num_envs * numel or src.dtype != dtype or \
                        (src.dim() > 1 and src.stride(-1) != 1) or not src.is_cuda:
                    raise _ffi.DtcError(f"add_transitions: field {i} has shape {tuple(src.shape)} / strides {src.stride()} / "
                                        f"{src.dtype} on {src.device}; expected [{self.num_envs}, {numel}] {dtype} rows with unit "
                                        "inner stride on the storage's device")
                it = items[i]
                it.src, it.dst = src.data_ptr(), base + s * step_bytes
                it.src_stride_bytes = src.stride(0) * src.element_size()
            to = None
            if time_outs is not None:
                to = time_outs.to(self.device)
                to = to.view(torch.uint8) if to.dtype == torch.bool else to.to(torch.uint8)
            ops.store_transition_items(items, len(srcs), f(tr.rewards).reshape(-1).contiguous(), f(tr.values).reshape(-1), to,
                                       gamma, self.rewards[s], self.num_envs)
        else:
            if time_outs is not None:
                tr.rewards = tr.rewards + gamma * torch.squeeze(tr.values * time_outs.unsqueeze(1).to(self.device), 1)
            self.observations[s].copy_(tr.observations)
            self.next_observations[s].copy_(tr.next_observations)
            self.privileged_observations[s].copy_(tr.privileged_observations)
            self.observation_histories[s].copy_(tr.observation_histories)
            self.actions[s].copy_(tr.actions)
            self.rewards[s].copy_(tr.rewards.view(-1, 1))
            self.dones[s].copy_(tr.dones.view(-1, 1))
            self.values[s].copy_(tr.values)
            self.actions_log_prob[s].copy_(tr.actions_log_prob.view(-1, 1))
            self.mu[s].copy_(tr.action_mean)
            self.base_vel[s].copy_(tr.base_vel)
            self.sigma[s].copy_(tr.action_sigma)
        self._save_hidden_states(tr.hidden_states)
        self.step += 1

    def _store_plan(self):
        dsts = (self.observations, self.next_observations, self.privileged_observations, self.observation_histories,
                self.actions, self.dones, self.values, self.actions_log_prob, self.mu, self.base_vel, self.sigma)
        plan = getattr(self, "_plan", None)
        ptrs = tuple(t.data_ptr() for t in dsts)
        if plan is None or plan["ptrs"] != ptrs:          # first use, or a buffer was replaced (.to(), re-assignment)
            items = (_ffi.DtcRowCopy * len(dsts))()
            meta = []
            for i, t in enumerate(dsts):
                row = t[0, 0].numel() * t.element_size()
                items[i].width_bytes = row
                meta.append((t.data_ptr(), t.stride(0) * t.element_size(), row, t[0, 0].numel(), t.dtype))
            plan = self._plan = dict(items=items, dst=meta, ptrs=ptrs)
        return plan

    def _save_hidden_states(self, hidden_states):
        """Record the recurrent states a rollout step started from (rollout_storage.py:118-132): `hidden_states` is the
        (actor, critic) pair, each a [layers, N, H] tensor (GRU) or an (h, c) tuple of them (LSTM); storage is one
        [T, layers, N, H] tensor per state tensor, allocated on first use."""
        if hidden_states is None or hidden_states == (None, None):
            return
        nets = [h if isinstance(h, tuple) else (h,) for h in hidden_states[:2]]
        if self.saved_hidden_states_a is None:
            T = self.num_transitions_per_env
            self.saved_hidden_states_a, self.saved_hidden_states_c = (
                [torch.zeros(T, *h.shape, device=self.device) for h in net] for net in nets)
        for saved, net in zip((self.saved_hidden_states_a, self.saved_hidden_states_c), nets):
            for dst, h in zip(saved, net):
                dst[self.step].copy_(h)

    def clear(self):
        self.step = 0

    def compute_returns(self, last_values, gamma, lam):
        """GAE(lambda) + advantage normalisation (rollout_storage.py:138-152).  Under torch.distributed
        the mean / unbiased std are those of the GLOBAL batch (two scalar all-reduces)."""
        T, N = self.num_transitions_per_env, self.num_envs
        if self._stats is None:
            self._stats = torch.zeros(4, dtype=torch.float64, device=self.device)
        ops.gae(self.rewards, self.values, self.dones, last_values.contiguous().float(), gamma, lam, self.returns,
                self.advantages, self._stats)
        world = dp.world_size()
        dp.allreduce_sum_(self._stats[0:1])
        count = float(T * N * world)
        ops.adv_sqdev(self.advantages, self._stats, count)
        dp.allreduce_sum_(self._stats[1:2])
        ops.adv_normalize(self.advantages, self._stats, count)

    def get_statistics(self):
        """(mean trajectory length, mean reward) of the stored rollout (rollout_storage.py:154-160).  The reference marks the
        last stored step as done IN the storage while it counts (`done = self.dones; done[-1] = 1`, an alias, not a copy);
        that side effect is replicated: a later compute_returns / recurrent generator sees the same dones as the reference's."""
        self.dones[-1] = 1
        _traj, _pos, lengths, _n = trajectory_index_map(self.dones)
        return lengths.float().mean(), self.rewards.mean()

    def flat(self, name):
        return getattr(self, name).flatten(0, 1)

    def mini_batch_generator(self, num_mini_batches, num_epochs=8, indices=None):
        """Yields the reference's 16-tuples (rollout_storage.py:213-214).  `indices` injects the permutation."""
        batch_size = self.num_envs * self.num_transitions_per_env
        mini_batch_size = batch_size // num_mini_batches
        if indices is None:
            indices = torch.randperm(num_mini_batches * mini_batch_size, requires_grad=False, device=self.device)
        names = ("observations", "privileged_observations", "observation_histories", "actions", "values",
                 "advantages", "returns", "actions_log_prob", "mu", "sigma", "base_vel", "next_observations", "rewards")
        flat = {k: self.flat(k) for k in names}
        for epoch in range(num_epochs):
            for i in range(num_mini_batches):
                batch_idx = indices[i * mini_batch_size:(i + 1) * mini_batch_size].contiguous()
                g = {k: ops.gather_rows(v, batch_idx) for k, v in flat.items()}
                yield (g["observations"], g["observations"], g["privileged_observations"],
                       g["observation_histories"], g["actions"], g["values"], g["advantages"], g["returns"],
                       g["actions_log_prob"], g["mu"], g["sigma"], g["base_vel"], g["next_observations"],
                       (None, None), None, g["rewards"])

    # ---- recurrent mini-batches (rollout_storage.py:217-267) ------------------------------------------------------
    # With more than one state tensor per network (LSTM: h and c) the reference hands the ACTOR's saved states to the critic
    # as well (rollout_storage.py:262).  "reference" reproduces that (an update then matches the reference's, pinned by
    # tests/golden/lstm.npz); "own" yields the critic's own states.  A GRU has one state tensor and is not affected.
    lstm_critic_states = "reference"

    def trajectory_layout(self, num_mini_batches):
        """Index data of the recurrent mini-batches of the stored rollout, computed once per update on the device:
        trajectories are numbered env-major (dtc_amd.utils.trajectory_index_map), so the trajectories of an env slice are
        a contiguous range `bounds[i]:bounds[i+1]`; `start_row[j]` = time-major storage row (t * N + env) of trajectory j's
        first step.  One host transfer (the num_mini_batches + 1 bounds) per call."""
        T, N = self.num_transitions_per_env, self.num_envs
        traj_id, _pos, lengths, n_traj = trajectory_index_map(self.dones)
        first = torch.cumsum(lengths, 0) - lengths                      # env-major index n * T + t of every trajectory start
        start_row = (first % T) * N + torch.div(first, T, rounding_mode="floor")
        mb = N // num_mini_batches
        slice_first = torch.arange(num_mini_batches, device=traj_id.device) * (mb * T)
        bounds = traj_id[slice_first].tolist() + [n_traj]
        return dict(bounds=bounds, start_row=start_row, lengths=lengths, n_traj=n_traj, mb=mb)

    def _states_at(self, saved, start_row):
        """[T, layers, N, H] saved states -> [layers, n_traj, H] at the trajectory starts (one row gather per state tensor)."""
        out = []
        for s in saved:
            T, L, N, H = s.shape
            out.append(s.transpose(1, 2).reshape(T * N, L, H).index_select(0, start_row).transpose(0, 1).contiguous())
        return out

    def reccurent_mini_batch_generator(self, num_mini_batches, num_epochs=8):
        """Yields the reference's 11-tuples: padded actor / critic observation trajectories of an env slice, the slice's
        rows of the flat quantities, the hidden states at its trajectory starts and the padding masks."""
        lay = self.trajectory_layout(num_mini_batches)
        padded_obs, masks = split_and_pad_trajectories(self.observations, self.dones)
        critic_src = self.privileged_observations if self.privileged_observations is not None else self.observations
        padded_critic = padded_obs if critic_src is self.observations else split_and_pad_trajectories(critic_src, self.dones)[0]
        hid_a_all = self._states_at(self.saved_hidden_states_a, lay["start_row"])
        if len(hid_a_all) > 1 and self.lstm_critic_states == "reference":
            hid_c_all = hid_a_all                                        # (sic) rollout_storage.py:262
        else:
            hid_c_all = self._states_at(self.saved_hidden_states_c, lay["start_row"])
        per_env = (self.actions, self.values, self.advantages, self.returns, self.actions_log_prob, self.mu, self.sigma)
        unwrap = lambda hs: hs[0] if len(hs) == 1 else hs
        for _epoch in range(num_epochs):
            for i in range(num_mini_batches):
                a, b = lay["bounds"][i], lay["bounds"][i + 1]
                envs = slice(i * lay["mb"], (i + 1) * lay["mb"])
                hid_a = unwrap([h[:, a:b].contiguous() for h in hid_a_all])
                hid_c = unwrap([h[:, a:b].contiguous() for h in hid_c_all])
                yield (padded_obs[:, a:b], padded_critic[:, a:b], *(t[:, envs] for t in per_env), (hid_a, hid_c), masks[:, a:b])
