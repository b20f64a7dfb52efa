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
from ..utils import split_and_pad_trajectories


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
            if not plan["checked"]:                       # shapes are fixed for the life of the storage: validate once
                for src, (base, step_bytes, width, numel) in zip(srcs, plan["dst"]):
                    assert src.shape[0] == self.num_envs and src[0].numel() == numel and (src.dim() == 1 or src.stride(-1) == 1)
                plan["checked"] = True
            for i, (src, (base, step_bytes, width, numel)) in enumerate(zip(srcs, plan["dst"])):
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
        plan = getattr(self, "_plan", None)
        if plan is None:
            dsts = (self.observations, self.next_observations, self.privileged_observations, self.observation_histories,
                    self.actions, self.dones, self.values, self.actions_log_prob, self.mu, self.base_vel, self.sigma)
            items = (_ffi.DtcRowCopy * len(dsts))()
            meta = []
            for i, t in enumerate(dsts):
                row = t[0, 0].numel() * t.element_size()
                items[i].width_bytes = row
                meta.append((t.data_ptr(), t.stride(0) * t.element_size(), row, t[0, 0].numel()))
            plan = self._plan = dict(items=items, dst=meta, checked=False)
        return plan

    def _save_hidden_states(self, hidden_states):
        if hidden_states is None or hidden_states == (None, None):
            return
        hid_a = hidden_states[0] if isinstance(hidden_states[0], tuple) else (hidden_states[0],)
        hid_c = hidden_states[1] if isinstance(hidden_states[1], tuple) else (hidden_states[1],)
        if self.saved_hidden_states_a is None:
            T = self.observations.shape[0]
            self.saved_hidden_states_a = [torch.zeros(T, *hid_a[i].shape, device=self.device) for i in range(len(hid_a))]
            self.saved_hidden_states_c = [torch.zeros(T, *hid_c[i].shape, device=self.device) for i in range(len(hid_c))]
        for i in range(len(hid_a)):
            self.saved_hidden_states_a[i][self.step].copy_(hid_a[i])
            self.saved_hidden_states_c[i][self.step].copy_(hid_c[i])

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
        done = self.dones
        done[-1] = 1
        flat_dones = done.permute(1, 0, 2).reshape(-1, 1)
        done_indices = torch.cat((flat_dones.new_tensor([-1], dtype=torch.int64), flat_dones.nonzero(as_tuple=False)[:, 0]))
        trajectory_lengths = (done_indices[1:] - done_indices[:-1])
        return trajectory_lengths.float().mean(), self.rewards.mean()

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

    # for RNNs only (rollout_storage.py:217-267)
    def reccurent_mini_batch_generator(self, num_mini_batches, num_epochs=8):
        padded_obs_trajectories, trajectory_masks = split_and_pad_trajectories(self.observations, self.dones)
        if self.privileged_observations is not None:
            padded_critic_obs_trajectories, _ = split_and_pad_trajectories(self.privileged_observations, self.dones)
        else:
            padded_critic_obs_trajectories = padded_obs_trajectories
        mini_batch_size = self.num_envs // num_mini_batches
        for ep in range(num_epochs):
            first_traj = 0
            for i in range(num_mini_batches):
                start, stop = i * mini_batch_size, (i + 1) * mini_batch_size
                dones = self.dones.squeeze(-1)
                last_was_done = torch.zeros_like(dones, dtype=torch.bool)
                last_was_done[1:] = dones[:-1]
                last_was_done[0] = True
                trajectories_batch_size = int(torch.sum(last_was_done[:, start:stop]))
                last_traj = first_traj + trajectories_batch_size
                masks_batch = trajectory_masks[:, first_traj:last_traj]
                obs_batch = padded_obs_trajectories[:, first_traj:last_traj]
                critic_obs_batch = padded_critic_obs_trajectories[:, first_traj:last_traj]
                sl = lambda t: t[:, start:stop]
                lwd = last_was_done.permute(1, 0)
                pick = lambda saved: [s.permute(2, 0, 1, 3)[lwd][first_traj:last_traj].transpose(1, 0).contiguous()
                                      for s in saved]
                hid_a_batch = pick(self.saved_hidden_states_a)
                hid_c_batch = pick(self.saved_hidden_states_c)
                hid_a_batch = hid_a_batch[0] if len(hid_a_batch) == 1 else hid_a_batch
                # (sic) rollout_storage.py:262 -- with more than one state tensor (LSTM: h and c) the reference hands the
                # ACTOR's saved states to the critic as well; kept, so that an update matches the reference's
                hid_c_batch = hid_c_batch[0] if len(hid_c_batch) == 1 else hid_a_batch
                yield (obs_batch, critic_obs_batch, sl(self.actions), sl(self.values), sl(self.advantages),
                       sl(self.returns), sl(self.actions_log_prob), sl(self.mu), sl(self.sigma),
                       (hid_a_batch, hid_c_batch), masks_batch)
                first_traj = last_traj
