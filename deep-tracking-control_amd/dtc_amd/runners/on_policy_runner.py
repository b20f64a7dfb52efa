"""OnPolicyRunner with the reference's constructor / learn / save / load / get_inference_policy
(rsl_rl/rsl_rl/runners/on_policy_runner.py:45-273).  The rollout loop and the checkpoint dictionary
layout are kept; TensorBoard is optional (logging is not part of the hot path)."""
import os
import statistics
import time
from collections import deque

import torch

from ..algorithms import PPO, RecurrentDecoderPPO
from ..env import HistoryWrapper
from ..modules import ActorCriticDecoder, ActorCriticDecoderRecurrent  # resolved by name from train_cfg

try:                                       # not installed in every image
    from torch.utils.tensorboard import SummaryWriter
except Exception:                          # pragma: no cover
    SummaryWriter = None

_POLICIES = {"ActorCriticDecoder": ActorCriticDecoder, "ActorCriticDecoderRecurrent": ActorCriticDecoderRecurrent}
_ALGORITHMS = {"PPO": PPO, "RecurrentDecoderPPO": RecurrentDecoderPPO}


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
        self.alg.init_storage(self.env.num_envs, self.num_steps_per_env, [self.env.num_obs],
                              [self.env.num_privileged_obs], [self.env.num_obs_history], [self.env.num_actions])
        self.log_dir = log_dir
        self.writer = None
        self.tot_timesteps = 0
        self.tot_time = 0
        self.current_learning_iteration = 0
        self.env.reset()

    def learn(self, num_learning_iterations, init_at_random_ep_len=False):
        if self.log_dir is not None and self.writer is None and SummaryWriter is not None:
            self.writer = SummaryWriter(log_dir=self.log_dir, flush_secs=10)
        if init_at_random_ep_len:
            self.env.episode_length_buf = torch.randint_like(self.env.episode_length_buf,
                                                             high=int(self.env.max_episode_length))
        obs_dict = self.env.get_observations()
        to = lambda t: t.to(self.device)
        obs, privileged_obs, obs_history = to(obs_dict["obs"]), to(obs_dict["privileged_obs"]), to(obs_dict["obs_history"])
        self.alg.actor_critic.train()
        ep_infos = []
        rewbuffer, lenbuffer = deque(maxlen=100), deque(maxlen=100)
        cur_reward_sum = torch.zeros(self.env.num_envs, dtype=torch.float, device=self.device)
        cur_episode_length = torch.zeros(self.env.num_envs, dtype=torch.float, device=self.device)
        rew_buf = self.env.get_reward_buf()
        tot_iter = self.current_learning_iteration + num_learning_iterations
        for it in range(self.current_learning_iteration, tot_iter):
            start = time.time()
            with torch.inference_mode():
                for i in range(self.num_steps_per_env):
                    actions = self.alg.act(obs, privileged_obs, obs_history, obs_dict['base_vel'], rew_buf)
                    obs_dict, rewards, dones, infos = self.env.step(actions)
                    obs, privileged_obs, obs_history = to(obs_dict["obs"]), to(obs_dict["privileged_obs"]), to(obs_dict["obs_history"])
                    rewards, dones = to(rewards), to(dones)
                    self.alg.process_env_step(rewards, dones, next_obs=obs_dict['obs'], infos=infos)
                    if self.log_dir is not None:
                        if 'episode' in infos:
                            ep_infos.append(infos['episode'])
                        cur_reward_sum += rewards
                        cur_episode_length += 1
                        new_ids = (dones > 0).nonzero(as_tuple=False)
                        rewbuffer.extend(cur_reward_sum[new_ids][:, 0].cpu().numpy().tolist())
                        lenbuffer.extend(cur_episode_length[new_ids][:, 0].cpu().numpy().tolist())
                        cur_reward_sum[new_ids] = 0
                        cur_episode_length[new_ids] = 0
                stop = time.time()
                collection_time = stop - start
                start = stop
                self.alg.compute_returns(obs, privileged_obs, obs_dict['base_vel'])
            (mean_value_loss, mean_surrogate_loss, mean_adaptation_module_loss, loss_decoder, mean_recons_loss,
             mean_vel_loss, mean_kld_loss) = self.alg.update()
            stop = time.time()
            learn_time = stop - start
            if self.log_dir is not None:
                self.log(locals())
            if it % self.save_interval == 0 and self.log_dir is not None:
                self.save(os.path.join(self.log_dir, 'model_{}.pt'.format(it)))
            ep_infos.clear()
        self.current_learning_iteration += num_learning_iterations
        if self.log_dir is not None:
            self.save(os.path.join(self.log_dir, 'model_{}.pt'.format(self.current_learning_iteration)))

    def log(self, locs, width=80, pad=35):
        self.tot_timesteps += self.num_steps_per_env * self.env.num_envs
        self.tot_time += locs['collection_time'] + locs['learn_time']
        fps = int(self.num_steps_per_env * self.env.num_envs / (locs['collection_time'] + locs['learn_time']))
        scalars = {
            'Loss/value_function': locs['mean_value_loss'], 'Loss/surrogate': locs['mean_surrogate_loss'],
            'Loss/recons_loss': locs['mean_recons_loss'], 'Loss/vel_loss': locs['mean_vel_loss'],
            'Loss/kld_loss': locs['mean_kld_loss'], 'Loss/learning_rate': self.alg.learning_rate,
            'Policy/mean_noise_std': float(self.alg.actor_critic.std.mean()), 'Perf/total_fps': fps,
            'Perf/collection time': locs['collection_time'], 'Perf/learning_time': locs['learn_time']}
        if len(locs['rewbuffer']) > 0:
            scalars['Train/mean_reward'] = statistics.mean(locs['rewbuffer'])
            scalars['Train/mean_episode_length'] = statistics.mean(locs['lenbuffer'])
        if self.writer is not None:
            for k, v in scalars.items():
                self.writer.add_scalar(k, v, locs['it'])
        head = f" Learning iteration {locs['it']}/{self.current_learning_iteration + locs['num_learning_iterations']} "
        lines = [head.center(width, ' ')] + [f"{k + ':':>{pad}} {v:.4f}" for k, v in scalars.items()]
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
