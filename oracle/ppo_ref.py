"""torch-CPU restatement of the reference's PPO update path (TEST INFRASTRUCTURE ONLY).

Restates, op for op on torch CPU tensors (the reference's own numeric backend):
  * `Vae` / `ActorCriticDecoder`      rsl_rl/rsl_rl/modules/actor_critic_decoder.py:91-302, 305-451, 540-551
  * `RolloutStorage` buffers, GAE     rsl_rl/rsl_rl/storage/rollout_storage.py:57-97, 138-152
  * `mini_batch_generator`            rsl_rl/rsl_rl/storage/rollout_storage.py:162-214
  * `PPO.update` (one mini-batch)     rsl_rl/rsl_rl/algorithms/ppo.py:189-338
  * `PPO.act` / `compute_returns`     rsl_rl/rsl_rl/algorithms/ppo.py:137-172

Pinned by tests/golden/ppo_*.npz (outputs of the imported reference on the same seeded
inputs; generator: tests/golden/make_golden.py).  The random draws the reference takes
from torch's global generator (`randperm`, two `randn_like` per mini-batch, one discarded
`Normal.sample`) are *inputs* here so that a GPU implementation can be fed the same numbers.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn as nn

NUM_OBS, NUM_PRIV, NUM_HIST, NUM_ACT = 53, 1389, 265, 12
N_HEIGHT, LATENT, TERRAIN_LATENT = 693, 16, 512


def _ortho(layer: nn.Linear, gain: float) -> nn.Linear:
    # actor_critic_decoder.py:268-272 -- orthogonal weight, zero bias
    nn.init.orthogonal_(layer.weight, gain)
    nn.init.constant_(layer.bias, 0.0)
    return layer


def _stack(dims, act_cls) -> nn.Sequential:
    """First Linear keeps PyTorch's default init, every later Linear is orthogonal(0.01);
    activation between layers, none on the output (actor_critic_decoder.py:98-113, 323-332)."""
    mods = [nn.Linear(dims[0], dims[1]), act_cls()]
    for i in range(1, len(dims) - 1):
        mods.append(_ortho(nn.Linear(dims[i], dims[i + 1]), 0.01))
        if i < len(dims) - 2:
            mods.append(act_cls())
    return nn.Sequential(*mods)


class RefVae(nn.Module):
    def __init__(self):
        super().__init__()
        self.cenet_encoder = _stack([NUM_HIST, 128, 64], nn.ReLU)
        self.latent_mu = _ortho(nn.Linear(64, 19), 0.01)
        self.latent_var = _ortho(nn.Linear(64, 16), 0.01)
        self.cenet_decoder = _stack([19 + TERRAIN_LATENT, 64, 128, NUM_OBS], nn.ReLU)
        self.terrain_encoder = _stack([N_HEIGHT, 512, 512, TERRAIN_LATENT], nn.ReLU)
        self.terrain_decoder = _stack([TERRAIN_LATENT, 512, 512, N_HEIGHT], nn.ReLU)
        self.memory_mlp = _stack([NUM_HIST + TERRAIN_LATENT, 256, 128, TERRAIN_LATENT], nn.ReLU)
        # The reference builds (and throws away) a 64->128->693 stack here
        # (actor_critic_decoder.py:209-228); it consumes init RNG, so do the same.
        _stack([64, 128, N_HEIGHT], nn.ReLU)
        self.gb_encoder = _stack([128, 128, 64], nn.ReLU)

    def cenet_forward(self, hist: torch.Tensor, eps: torch.Tensor):
        """actor_critic_decoder.py:286-302 with the reparameterisation noise injected."""
        e = self.cenet_encoder(hist)
        lv = self.latent_var(e)
        mu = self.latent_mu(e)
        mean = lv.mean()
        std = lv.std()
        thr = 2 * std
        outliers = (lv < (mean - thr)) | (lv > (mean + thr))
        median = lv[~outliers].median()
        with torch.no_grad():     # bookkeeping for the parity tests (which element IS the median)
            self.last_outliers, self.last_median = int(outliers.sum()), float(median)
            self.last_median_index = int(((lv == median) & ~outliers).flatten().nonzero()[0])
            self.last_outlier_mask = outliers.detach().clone()
        lv[outliers] = median
        self.last_logvar = lv.detach().clone()      # after the replacement (what the reparameterisation saw)
        z = eps * torch.exp(0.5 * lv) + mu[:, 3:]
        return mu, lv, z


class RefActorCriticDecoder(nn.Module):
    is_recurrent = False

    def __init__(self, num_obs=NUM_OBS, num_critic_obs=NUM_PRIV, num_actions=NUM_ACT):
        super().__init__()
        self.vae = RefVae()
        self.actor_body = _stack([num_obs + 16 + 3 + TERRAIN_LATENT, 512, 256, 128, num_actions], nn.ELU)
        self.critic_body = _stack([N_HEIGHT + num_obs + 3 + 3, 512, 256, 128, 1], nn.ELU)
        self.std = nn.Parameter(torch.ones(num_actions))

    # actor_critic_decoder.py:409-437
    def policy_mean(self, obs, hist, priv, eps):
        mu, lv, z = self.vae.cenet_forward(hist, eps)
        l_t = self.vae.terrain_encoder(priv[:, :N_HEIGHT])
        return self.actor_body(torch.cat((obs, z, mu[:, :3], l_t), dim=-1))

    # actor_critic_decoder.py:504-538 (deployment path; `memory_mlp` never receives a gradient in PPO.update)
    def act_teacher(self, obs, hist, priv):
        vae = self.vae
        latent = vae.latent_mu(vae.cenet_encoder(hist))
        l_t = vae.terrain_encoder(priv[:, :N_HEIGHT])
        b_t1 = vae.memory_mlp(torch.cat((hist, l_t), dim=-1))
        b_t = b_t1 + torch.mul(l_t, b_t1)
        return self.actor_body(torch.cat((obs, latent[:, 3:], latent[:, :3], b_t), dim=-1))

    # actor_critic_decoder.py:459-502 with its dangling names resolved (cenet_encoder / latent_mu -> vae.*, actor_student -> actor_body;
    # the exporter side effect dropped: its output is discarded there): mean action from latent_mu and a caller-supplied lidar latent
    def act_student(self, obs, hist, priv, lidar_latent):
        latent = self.vae.latent_mu(self.vae.cenet_encoder(hist))
        return self.actor_body(torch.cat((obs, latent[:, 3:], latent[:, :3], lidar_latent), dim=-1))

    # actor_critic_decoder.py:404-407
    @staticmethod
    def adapt_bootstrap_probability(rewards):
        cv = torch.std(rewards) / torch.mean(rewards)
        return (1 - torch.tanh(cv)).item()

    # actor_critic_decoder.py:540-551
    def evaluate(self, obs, priv, base_vel):
        return self.critic_body(torch.cat((obs, base_vel, priv[:, 693:696], priv[:, 696:]), dim=-1))


def fill_parameters_(module: nn.Module, seed: int, scale: float = 1.0):
    """Deterministic, init-order-independent fill used by the parity fixtures: every tensor
    of `state_dict()` (in key order) gets randn/sqrt(fan_in) from its own seeded generator;
    biases get 0.1*randn; `std` gets 1 + 0.1*randn clipped to [0.5, 1.5]."""
    with torch.no_grad():
        for i, (k, v) in enumerate(module.state_dict().items()):
            g = torch.Generator().manual_seed(seed * 1000 + i)
            r = torch.randn(v.shape, generator=g)
            if k.endswith("std"):
                v.copy_((1.0 + 0.1 * r).clamp(0.5, 1.5))
            elif v.dim() == 2:
                v.copy_(r * (scale / math.sqrt(v.shape[1])))
            else:
                v.copy_(0.1 * r)
    return module


class RefStorage:
    """rollout_storage.py:57-97 buffers ([T,N,d], time-major) + GAE + flat mini-batch gather."""

    FIELDS = dict(observations=NUM_OBS, next_observations=NUM_OBS, privileged_observations=NUM_PRIV,
                  observation_histories=NUM_HIST, rewards=1, actions=NUM_ACT, actions_log_prob=1,
                  values=1, returns=1, advantages=1, mu=NUM_ACT, sigma=NUM_ACT, base_vel=3)

    def __init__(self, num_envs, num_steps):
        self.num_envs, self.num_steps = num_envs, num_steps
        for k, d in self.FIELDS.items():
            setattr(self, k, torch.zeros(num_steps, num_envs, d))
        self.dones = torch.zeros(num_steps, num_envs, 1, dtype=torch.uint8)

    def compute_returns(self, last_values, gamma, lam):
        # rollout_storage.py:138-152
        advantage = 0
        T = self.num_steps
        for step in reversed(range(T)):
            next_values = last_values if step == T - 1 else self.values[step + 1]
            nnt = 1.0 - self.dones[step].float()
            delta = self.rewards[step] + nnt * gamma * next_values - self.values[step]
            advantage = delta + nnt * gamma * lam * advantage
            self.returns[step] = advantage + self.values[step]
        self.advantages = self.returns - self.values
        self.advantages = (self.advantages - self.advantages.mean()) / (self.advantages.std() + 1e-8)

    def gather(self, idx):
        """One mini-batch of rollout_storage.py:188-214 (16-tuple order, Appendix A.5)."""
        f = lambda t: t.flatten(0, 1)[idx]
        return (f(self.observations), f(self.observations), f(self.privileged_observations),
                f(self.observation_histories), f(self.actions), f(self.values), f(self.advantages),
                f(self.returns), f(self.actions_log_prob), f(self.mu), f(self.sigma), f(self.base_vel),
                f(self.next_observations), (None, None), None, f(self.rewards))


@dataclass
class StepRecord:
    recons: float = 0.0
    vel: float = 0.0
    kld: float = 0.0
    height: float = 0.0
    vae_gnorm: float = 0.0
    surrogate: float = 0.0
    value: float = 0.0
    entropy: float = 0.0
    kl_mean: float = 0.0
    lr: float = 0.0
    gnorm: float = 0.0
    extra: dict = field(default_factory=dict)


class RefPPO:
    """ppo.py:42-357 for ActorCriticDecoder; defaults are PPO.__init__'s (ppo.py:45-61)."""

    def __init__(self, actor_critic, num_learning_epochs=5, num_mini_batches=4, clip_param=0.2, gamma=0.99,
                 lam=0.95, value_loss_coef=1.0, entropy_coef=0.01, learning_rate=5.e-4, max_grad_norm=1.0,
                 use_clipped_value_loss=True, schedule="adaptive", desired_kl=0.01):
        self.actor_critic = actor_critic
        self.optimizer = torch.optim.Adam(actor_critic.parameters(), lr=learning_rate)   # ppo.py:78
        self.vae_optimizer = torch.optim.Adam(actor_critic.vae.parameters(), lr=5.e-4)   # ppo.py:79
        self.learning_rate = learning_rate
        self.num_learning_epochs, self.num_mini_batches = num_learning_epochs, num_mini_batches
        self.clip_param, self.gamma, self.lam = clip_param, gamma, lam
        self.value_loss_coef, self.entropy_coef = value_loss_coef, entropy_coef
        self.max_grad_norm, self.use_clipped_value_loss = max_grad_norm, use_clipped_value_loss
        self.schedule, self.desired_kl = schedule, desired_kl
        self.storage = None
        self.capture_grads = False      # tests: keep the pre-clip gradients of both steps in StepRecord.extra
        self.relu_masks = {}            # tests: name -> (output > 0) of every ReLU of the VAE, last forward
        for name, mod in actor_critic.vae.named_modules():
            if isinstance(mod, nn.ReLU):
                mod.register_forward_hook(lambda m, i, o, name=name: self.relu_masks.__setitem__(name, (o > 0).detach()))

    def init_storage(self, num_envs, num_steps):
        self.storage = RefStorage(num_envs, num_steps)

    # ---- rollout side (ppo.py:137-172) -------------------------------------------------
    @torch.no_grad()
    def act(self, obs, priv, hist, base_vel, eps, noise):
        ac = self.actor_critic
        mean = ac.policy_mean(obs, hist, priv, eps)
        sigma = mean * 0. + ac.std
        actions = mean + sigma * noise
        values = ac.evaluate(obs, priv, base_vel)
        logp = torch.distributions.Normal(mean, sigma).log_prob(actions).sum(dim=-1)
        return actions, values, logp, mean, sigma

    @torch.no_grad()
    def compute_returns(self, last_obs, last_priv, last_base_vel):
        last_values = self.actor_critic.evaluate(last_obs, last_priv, last_base_vel)
        self.storage.compute_returns(last_values, self.gamma, self.lam)

    # ---- one mini-batch of PPO.update (ppo.py:189-338) ---------------------------------
    def step(self, idx, eps1, eps2) -> StepRecord:
        rec = StepRecord()
        self.vae_step(idx, eps1, rec)
        self.ppo_step(idx, eps2, rec)
        return rec

    def vae_step(self, idx, eps1, rec: StepRecord) -> StepRecord:
        """First half of a mini-batch (ppo.py:197-258)."""
        ac, vae = self.actor_critic, self.actor_critic.vae
        (obs, critic_obs, priv, hist, actions, target_values, advantages, returns, old_logp, old_mu,
         old_sigma, base_vel, next_obs, _, _, _) = self.storage.gather(idx)

        mu, lv, z = vae.cenet_forward(hist, eps1)
        l_t = vae.terrain_encoder(priv[:, :N_HEIGHT])
        recons = vae.cenet_decoder(torch.cat([z, mu[:, :3], l_t], dim=1))
        recons_loss = torch.pow(recons - next_obs, 2).mean(-1).mean()
        height_recon = vae.terrain_decoder(l_t)
        height_loss = nn.functional.mse_loss(height_recon, priv[..., 696:])
        vel_loss = nn.functional.mse_loss(mu[:, :3], base_vel)
        kld_loss = torch.mean(-0.5 * torch.sum(1 + lv - mu[:, 3:].pow(2) - lv.exp(), dim=1))
        vae_loss = recons_loss + vel_loss + 4 * kld_loss + height_loss
        self.vae_optimizer.zero_grad()
        vae_loss.backward()
        if self.capture_grads:
            rec.extra["vae_grads"] = {k: p.grad.clone() for k, p in ac.named_parameters()
                                      if p.grad is not None and k.startswith("vae.")}
        rec.vae_gnorm = float(nn.utils.clip_grad_norm_(vae.parameters(), self.max_grad_norm))
        self.vae_optimizer.step()
        rec.recons, rec.vel, rec.kld, rec.height = (recons_loss.item(), vel_loss.item(), kld_loss.item(),
                                                    height_loss.item())
        return rec

    def ppo_step(self, idx, eps2, rec: StepRecord) -> StepRecord:
        """Second half of a mini-batch (ppo.py:265-338); the VAE weights are the freshly updated ones."""
        ac, vae = self.actor_critic, self.actor_critic.vae
        (obs, critic_obs, priv, hist, actions, target_values, advantages, returns, old_logp, old_mu,
         old_sigma, base_vel, next_obs, _, _, _) = self.storage.gather(idx)
        mean = ac.policy_mean(obs, hist, priv, eps2)
        value = ac.evaluate(critic_obs, priv, base_vel)
        return self._ppo_tail(mean, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, rec)

    def _ppo_tail(self, mean, value, actions, target_values, advantages, returns, old_logp, old_mu, old_sigma, rec):
        """ppo.py:288-338: log-prob / entropy / KL-adaptive learning rate / losses / backward / clip / Adam."""
        ac = self.actor_critic
        dist = torch.distributions.Normal(mean, mean * 0. + ac.std)
        logp = dist.log_prob(actions).sum(dim=-1)
        sigma = dist.stddev
        entropy = dist.entropy().sum(dim=-1)

        if self.desired_kl is not None and self.schedule == 'adaptive':
            with torch.inference_mode():
                kl = torch.sum(torch.log(sigma / old_sigma + 1.e-5)
                               + (torch.square(old_sigma) + torch.square(old_mu - mean))
                               / (2.0 * torch.square(sigma)) - 0.5, axis=-1)
                kl_mean = torch.mean(kl)
                if kl_mean > self.desired_kl * 2.0:
                    self.learning_rate = max(1e-5, self.learning_rate / 1.5)
                elif kl_mean < self.desired_kl / 2.0 and kl_mean > 0.0:
                    self.learning_rate = min(1e-2, self.learning_rate * 1.5)
                for g in self.optimizer.param_groups:
                    g['lr'] = self.learning_rate
                rec.kl_mean = kl_mean.item()

        ratio = torch.exp(logp - torch.squeeze(old_logp))
        adv = torch.squeeze(advantages)
        surrogate = -adv * ratio
        surrogate_clipped = -adv * torch.clamp(ratio, 1.0 - self.clip_param, 1.0 + self.clip_param)
        surrogate_loss = torch.max(surrogate, surrogate_clipped).mean()
        if self.use_clipped_value_loss:
            value_clipped = target_values + (value - target_values).clamp(-self.clip_param, self.clip_param)
            value_loss = torch.max((value - returns).pow(2), (value_clipped - returns).pow(2)).mean()
        else:
            value_loss = (returns - value).pow(2).mean()
        loss = surrogate_loss + self.value_loss_coef * value_loss - self.entropy_coef * entropy.mean()
        self.optimizer.zero_grad()
        loss.backward()
        if self.capture_grads:
            rec.extra["grads"] = {k: p.grad.clone() for k, p in ac.named_parameters() if p.grad is not None}
        rec.gnorm = float(nn.utils.clip_grad_norm_(ac.parameters(), self.max_grad_norm))
        self.optimizer.step()
        rec.surrogate, rec.value, rec.entropy = surrogate_loss.item(), value_loss.item(), entropy.mean().item()
        rec.lr = self.learning_rate
        return rec

    def update(self, perm, eps1, eps2, record=None):
        """Full PPO.update: `perm` [num_mini_batches*mb] int64 (one per update, reused by all
        epochs -- rollout_storage.py:165), eps1/eps2 [steps, mb, 16]."""
        mb = (self.storage.num_envs * self.storage.num_steps) // self.num_mini_batches
        sums = dict(value=0.0, surrogate=0.0, recons=0.0, vel=0.0, kld=0.0)
        k = 0
        for _ in range(self.num_learning_epochs):
            for i in range(self.num_mini_batches):
                r = self.step(perm[i * mb:(i + 1) * mb], eps1[k], eps2[k])
                if record is not None:
                    record.append(r)
                for key in sums:
                    sums[key] += getattr(r, key)
                k += 1
        n = self.num_learning_epochs * self.num_mini_batches
        return (sums['value'] / n, sums['surrogate'] / n, 0.0, 0, sums['recons'] / n, sums['vel'] / n,
                sums['kld'] / n)
