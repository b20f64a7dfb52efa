"""OnPolicyRunner with the reference's constructor / learn / save / load / get_inference_policy
(rsl_rl/rsl_rl/runners/on_policy_runner.py:45-273).  The rollout loop and the checkpoint dictionary
layout are kept; TensorBoard is optional (logging is not part of the hot path)."""
import os
import time

import torch

from ..algorithms import PPO, RecurrentDecoderPPO, RecurrentPPO
from ..env import HistoryWrapper
from ..modules import ActorCritic, ActorCriticDecoder, ActorCriticDecoderRecurrent, ActorCriticRecurrent  # resolved by name from train_cfg

try:                                       # not installed in every image
    from torch.utils.tensorboard import SummaryWriter
except Exception:                          # pragma: no cover
    SummaryWriter = None

# on_policy_runner.py:38-42, 60, 67 resolve the classes with eval(): the same names are reachable here.  `PPO` needs a
# model with a `.vae` (ppo.py:79) exactly as in the reference; the recurrent actor-critic of BASELINE configs[2] trains
# with `RecurrentPPO` (the upstream PPO step the fork dropped), the composite with `RecurrentDecoderPPO`.
_POLICIES = {"ActorCritic": ActorCritic, "ActorCriticRecurrent": ActorCriticRecurrent, "ActorCriticDecoder": ActorCriticDecoder,
             "ActorCriticDecoderRecurrent": ActorCriticDecoderRecurrent}
_ALGORITHMS = {"PPO": PPO, "RecurrentPPO": RecurrentPPO, "RecurrentDecoderPPO": RecurrentDecoderPPO}


class _EpisodeTracker:
    """Return and length of the episode in flight per env; finished episodes go into a device-side ring of the most
    recent `keep` (the numbers the reference's logger averages) -- no host round trip per env step, one at log time."""

    def __init__(self, num_envs, device, keep=100):
        self.keep = keep
        self.ret = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self.length = torch.zeros(num_envs, dtype=torch.float32, device=device)
        self.ring = torch.zeros(keep + 1, 2, dtype=torch.float32, device=device)     # slot `keep` absorbs the non-finished envs
        self.finished = torch.zeros((), dtype=torch.long, device=device)

    def step(self, rewards, dones):
        self.ret += rewards
        self.length += 1
        done = dones > 0
        order = torch.cumsum(done, 0) - 1                                   # 0, 1, ... among the envs finishing now
        slot = torch.where(done, (self.finished + order) % self.keep, torch.full_like(order, self.keep))
        self.ring[slot] = torch.stack((self.ret, self.length), dim=1)
        self.finished += done.sum()
        keep_going = (~done).to(self.ret.dtype)
        self.ret *= keep_going
        self.length *= keep_going

    def means(self):
        n = min(int(self.finished), self.keep)
        if n == 0:
            return None
        m = self.ring[:n].mean(dim=0)
        return float(m[0]), float(m[1])


class OnPolicyRunner:
    def __init__(self, env, train_cfg, log_dir=None, device='cpu'):
        self.cfg = train_cfg["runner"]
        self.alg_cfg = train_cfg["algorithm"]
        self.policy_cfg = train_cfg["policy"]
        self.device = device
        self.env = HistoryWrapper(env)
        num_critic_obs = self.env.num_privileged_obs if self.env.num_privileged_obs is not None else self.env.num_obs
        actor_critic_class = _POLICIES[self.cfg["policy_class_name"]]
        actor_critic = actor_critic_class(self.env.num_obs, num_critic_obs, self.env.num_actions,
                                          **self.policy_cfg).to(self.device)
        alg_class = _ALGORITHMS[self.cfg["algorithm_class_name"]]
        self.alg = alg_class(actor_critic, device=self.device, **self.alg_cfg)
        self.num_steps_per_env = self.cfg["num_steps_per_env"]
        self.save_interval = self.cfg["save_interval"]
        # upstream-style trainers (actor obs / critic obs only) take no observation history, base velocity or reward buffer
        self._plain = isinstance(self.alg, RecurrentPPO)
        if self._plain:
            self.alg.init_storage(self.env.num_envs, self.num_steps_per_env, [self.env.num_obs], [num_critic_obs],
                                  [self.env.num_actions])
        else:
            self.alg.init_storage(self.env.num_envs, self.num_steps_per_env, [self.env.num_obs],
                                  [self.env.num_privileged_obs], [self.env.num_obs_history], [self.env.num_actions])
        self.log_dir = log_dir
        self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.env.reset()

    # ---------------------------------------------------------------- training loop
    def _observe(self, obs_dict):
        dev = self.device
        return obs_dict["obs"].to(dev), obs_dict["privileged_obs"].to(dev), obs_dict["obs_history"].to(dev)

    def _collect(self, state, tracker, ep_infos):
        """One rollout of `num_steps_per_env` env steps into the storage, then the bootstrap values / returns."""
        obs_dict, rew_buf = state["obs_dict"], state["rew_buf"]
        obs, priv, hist = self._observe(obs_dict)
        with torch.inference_mode():
            for _ in range(self.num_steps_per_env):
                actions = self.alg.act(obs, priv) if self._plain else self.alg.act(obs, priv, hist, obs_dict['base_vel'], rew_buf)
                obs_dict, rewards, dones, infos = self.env.step(actions)
                obs, priv, hist = self._observe(obs_dict)
                rewards, dones = rewards.to(self.device), dones.to(self.device)
                self.alg.process_env_step(rewards, dones, next_obs=obs_dict['obs'], infos=infos)
                if tracker is not None:
                    tracker.step(rewards, dones)
                    if 'episode' in infos:
                        ep_infos.append(infos['episode'])
            if self._plain:
                self.alg.compute_returns(priv)
            else:
                self.alg.compute_returns(obs, priv, obs_dict['base_vel'])
        state["obs_dict"] = obs_dict

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        logging = self.log_dir is not None
        if logging and self.writer is None and SummaryWriter is not None:
            self.writer = SummaryWriter(log_dir=self.log_dir, flush_secs=10)
        if init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf,
                                                             high=int(self.env.max_episode_length))
        self.alg.actor_critic.train()
        state = dict(obs_dict=self.env.get_observations(), rew_buf=self.env.get_reward_buf())
        tracker = _EpisodeTracker(self.env.num_envs, self.device) if logging else None
        ep_infos = []
        first, last = self.current_learning_iteration, self.current_learning_iteration + num_learning_iterations
        for it in range(first, last):
            t0 = time.time()
            self._collect(state, tracker, ep_infos)
            t1 = time.time()
            losses = self.alg.update()           # (value, surrogate, adaptation, decoder, recons, vel, kld) means
            t2 = time.time()
            if logging:
                self.log(it, last, losses, collection_time=t1 - t0, learn_time=t2 - t1, tracker=tracker)
                if it % self.save_interval == 0:
                    self.save(os.path.join(self.log_dir, f'model_{it}.pt'))
            ep_infos.clear()
        self.current_learning_iteration = last
        if logging:
            self.save(os.path.join(self.log_dir, f'model_{last}.pt'))

    def log(self, it, last, losses, collection_time, learn_time, tracker, width=80, pad=35):
        value_loss, surrogate_loss, _adaptation, _decoder, recons_loss, vel_loss, kld_loss = tuple(losses) + (0.0,) * (7 - len(losses))
        steps = self.num_steps_per_env * self.env.num_envs
        self.tot_timesteps += steps
        self.tot_time += collection_time + learn_time
        scalars = {
            'Loss/value_function': value_loss, 'Loss/surrogate': surrogate_loss, 'Loss/recons_loss': recons_loss,
            'Loss/vel_loss': vel_loss, 'Loss/kld_loss': kld_loss, 'Loss/learning_rate': self.alg.learning_rate,
            'Policy/mean_noise_std': float(self.alg.actor_critic.std.mean()),
            'Perf/total_fps': int(steps / (collection_time + learn_time)),
            'Perf/collection time': collection_time, 'Perf/learning_time': learn_time}
        finished = tracker.means()
        if finished is not None:
            scalars['Train/mean_reward'], scalars['Train/mean_episode_length'] = finished
        if self.writer is not None:
            for k, v in scalars.items():
                self.writer.add_scalar(k, v, it)
        lines = [f" Learning iteration {it}/{last} ".center(width, ' ')]
        lines += [f"{k + ':':>{pad}} {v:.4f}" for k, v in scalars.items()]
        print("#" * width + "\n" + "\n".join(lines))

    def save(self, path, infos=None):
        torch.save({'model_state_dict': self.alg.actor_critic.state_dict(),
                    'optimizer_state_dict': self.alg.optimizer.state_dict(),
                    'iter': self.current_learning_iteration, 'infos': infos}, path)

    def load(self, path, load_optimizer=True):
        loaded_dict = torch.load(path, map_location="cpu")
        self.alg.actor_critic.load_state_dict(loaded_dict['model_state_dict'])
        if load_optimizer:
            self.alg.optimizer.load_state_dict(loaded_dict['optimizer_state_dict'])
            self.alg.learning_rate = self.alg.optimizer.param_groups[0]['lr']
        self.current_learning_iteration = loaded_dict['iter']
        return loaded_dict['infos']

    def get_inference_policy(self, env_t=None, device=None):
        self.alg.actor_critic.eval()
        if device is not None:
            self.alg.actor_critic.to(device)
        if env_t == True:                      # noqa: E712  (on_policy_runner.py:269-272)
            return self.alg.actor_critic.act_expert
        return self.alg.actor_critic.act_inference
