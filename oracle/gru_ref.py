"""torch-CPU restatement of the recurrent (GRU / LSTM) actor-critic path (TEST INFRASTRUCTURE ONLY).

Restates rsl_rl/rsl_rl/modules/actor_critic.py:38-155, actor_critic_recurrent.py:40-116 (`ActorCriticRecurrent`,
`Memory` around torch.nn.GRU), rsl_rl/rsl_rl/utils/utils.py:33-70 (split/pad/unpad) and the recurrent mini-batch
of rollout_storage.py:217-267.  The reference's own `PPO` cannot train this model at this commit,
so the training step is the upstream rsl_rl PPO step = ppo.py:288-335 without the VAE block (SURVEY.md §8a
"Config 3"); that loss block is the one already pinned by tests/golden/ppo.npz.  The modules / padding are pinned
by tests/golden/gru.npz and tests/golden/lstm.npz (outputs of the imported reference classes; the LSTM fixture uses the
reference's default `rnn_type='lstm'` with two layers).
"""
from __future__ import annotations

import torch
import torch.nn as nn


def _mlp(in_dim, hidden, out_dim):
    dims = [in_dim] + list(hidden) + [out_dim]
    mods = []
    for i in range(len(dims) - 1):
        mods.append(nn.Linear(dims[i], dims[i + 1]))
        if i < len(dims) - 2:
            mods.append(nn.ELU())
    return nn.Sequential(*mods)


class RefMemory(nn.Module):
    def __init__(self, input_size, hidden_size, rnn_type='gru', num_layers=1):
        super().__init__()
        rnn_cls = nn.GRU if rnn_type.lower() == 'gru' else nn.LSTM          # actor_critic_recurrent.py:96-97
        self.rnn = rnn_cls(input_size=input_size, hidden_size=hidden_size, num_layers=num_layers)


class RefActorCriticRecurrent(nn.Module):
    def __init__(self, num_actor_obs=53, num_critic_obs=1389, num_actions=12, hidden=(512, 256, 128), rnn_hidden=512,
                 rnn_type='gru', num_layers=1):
        super().__init__()
        self.actor = _mlp(rnn_hidden, hidden, num_actions)
        self.critic = _mlp(rnn_hidden, hidden, 1)
        self.std = nn.Parameter(torch.ones(num_actions))
        self.memory_a = RefMemory(num_actor_obs, rnn_hidden, rnn_type, num_layers)
        self.memory_c = RefMemory(num_critic_obs, rnn_hidden, rnn_type, num_layers)


def split_and_pad(tensor, dones):
    """utils.py:33-64 (padded to the longest trajectory; the tests keep one env without resets)."""
    dones = dones.clone()
    dones[-1] = 1
    flat = dones.transpose(1, 0).reshape(-1, 1)
    ends = torch.cat((flat.new_tensor([-1], dtype=torch.int64), flat.nonzero()[:, 0]))
    lengths = ends[1:] - ends[:-1]
    trajs = torch.split(tensor.transpose(1, 0).flatten(0, 1), lengths.tolist())
    padded = nn.utils.rnn.pad_sequence(trajs)
    masks = lengths > torch.arange(0, tensor.shape[0]).unsqueeze(1)
    return padded, masks


def unpad(traj, masks):
    return traj.transpose(1, 0)[masks.transpose(1, 0)].view(-1, traj.shape[0], traj.shape[-1]).transpose(1, 0)


def recurrent_batches(st, hid_a, hid_c, num_mini_batches):
    """rollout_storage.py:217-267 for one epoch.  st: oracle RefStorage-like with [T,N,.] tensors;
    hid_a/hid_c: saved hidden states [T, L, N, H] (state BEFORE each step); a tuple (h, c) of two such for an LSTM."""
    obs_p, masks = split_and_pad(st.observations, st.dones)
    cobs_p, _ = split_and_pad(st.privileged_observations, st.dones)
    N = st.observations.shape[1]
    mb = N // num_mini_batches
    dones = st.dones.squeeze(-1)
    lwd = torch.zeros_like(dones, dtype=torch.bool)
    lwd[1:] = dones[:-1].bool()
    lwd[0] = True
    first = 0
    for i in range(num_mini_batches):
        a, b = i * mb, (i + 1) * mb
        n_traj = int(lwd[:, a:b].sum())
        last = first + n_traj
        pick1 = lambda h: h.permute(2, 0, 1, 3)[lwd.permute(1, 0)][first:last].transpose(1, 0).contiguous()
        pick = lambda h: tuple(pick1(x) for x in h) if isinstance(h, (tuple, list)) else pick1(h)     # LSTM: (h, c)
        # rollout_storage.py:261-262: `hid_c_batch = hid_c_batch[0] if len(hid_c_batch)==1 else hid_a_batch` (sic) -- for an
        # LSTM the critic receives the actor's saved states
        ha, hc = pick(hid_a), pick(hid_c)
        yield dict(obs=obs_p[:, first:last], cobs=cobs_p[:, first:last], masks=masks[:, first:last],
                   hid_a=ha, hid_c=(ha if isinstance(hid_c, (tuple, list)) else hc), sl=slice(a, b))
        first = last


class RefRecurrentPPO:
    """Upstream rsl_rl PPO step on the recurrent model (ppo.py:288-335 semantics)."""

    def __init__(self, ac, clip_param=0.2, value_loss_coef=1.0, entropy_coef=0.01, learning_rate=1e-3,
                 max_grad_norm=1.0, use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01):
        self.ac = ac
        self.optimizer = torch.optim.Adam(ac.parameters(), lr=learning_rate)
        self.learning_rate = learning_rate
        self.clip_param, self.value_loss_coef, self.entropy_coef = clip_param, value_loss_coef, entropy_coef
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.schedule, self.desired_kl = schedule, desired_kl
        self.capture_grads = False

    def forward(self, batch):
        ac = self.ac
        out_a, _ = ac.memory_a.rnn(batch["obs"], batch["hid_a"])
        mean = ac.actor(unpad(out_a, batch["masks"]))
        out_c, _ = ac.memory_c.rnn(batch["cobs"], batch["hid_c"])
        value = ac.critic(unpad(out_c, batch["masks"]))
        return mean, value

    def step(self, st, batch):
        ac, sl = self.ac, batch["sl"]
        mean, value = self.forward(batch)
        dist = torch.distributions.Normal(mean, mean * 0. + ac.std)
        actions, old_logp = st.actions[:, sl], st.actions_log_prob[:, sl]
        logp = dist.log_prob(actions).sum(dim=-1)
        sigma, entropy = dist.stddev, dist.entropy().sum(dim=-1)
        rec = {}
        if self.desired_kl is not None and self.schedule == 'adaptive':
            with torch.inference_mode():
                kl = torch.sum(torch.log(sigma / st.sigma[:, sl] + 1.e-5)
                               + (torch.square(st.sigma[:, sl]) + torch.square(st.mu[:, sl] - mean))
                               / (2.0 * torch.square(sigma)) - 0.5, axis=-1)
                kl_mean = torch.mean(kl)
                if kl_mean > self.desired_kl * 2.0:
                    self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                    self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                for g in self.optimizer.param_groups:
                    g['lr'] = self.learning_rate
                rec["kl_mean"] = kl_mean.item()
        ratio = torch.exp(logp - torch.squeeze(old_logp))
        adv = torch.squeeze(st.advantages[:, sl])
        surrogate_loss = torch.max(-adv * ratio, -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)).mean()
        tv, ret = st.values[:, sl], st.returns[:, sl]
        if self.use_clipped_value_loss:
            vc = tv + (value - tv).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - ret).pow(2), (vc - ret).pow(2)).mean()
        else:
            value_loss = (ret - value).pow(2).mean()
        loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
        self.optimizer.zero_grad()
        loss.backward()
        if self.capture_grads:
            rec["grads"] = {k: p.grad.clone() for k, p in ac.named_parameters() if p.grad is not None}
        rec["gnorm"] = float(nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm))
        self.optimizer.step()
        rec.update(surrogate=surrogate_loss.item(), value=value_loss.item(), entropy=entropy.mean().item(),
                   lr=self.learning_rate, mean=mean.detach(), value_out=value.detach())
        return rec
