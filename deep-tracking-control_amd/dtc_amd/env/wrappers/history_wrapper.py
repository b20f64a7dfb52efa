"""HistoryWrapper: dict-observation packaging + 5-step observation history roll
(rsl_rl/rsl_rl/env/wrappers/history_wrapper.py:6-53).  No gym dependency: attribute access falls
through to the wrapped env."""
import torch


class HistoryWrapper:
    def __init__(self, env):
        self.env = env
        self.obs_history_length = self.env.cfg.env.num_observation_history
        self.num_obs_history = self.obs_history_length * self.env.num_obs
        self.obs_history = torch.zeros(self.env.num_envs, self.num_obs_history, dtype=torch.float,
                                       device=self.env.device, requires_grad=False)

    def __getattr__(self, name):
        return getattr(self.__dict__["env"], name)

    def __setattr__(self, name, value):
        if name in ("env", "obs_history_length", "num_obs_history", "obs_history") or "env" not in self.__dict__:
            object.__setattr__(self, name, value)
        elif hasattr(self.__dict__["env"], name) and name not in self.__dict__:
            setattr(self.__dict__["env"], name, value)      # e.g. runner sets env.episode_length_buf
        else:
            object.__setattr__(self, name, value)

    def _roll(self, obs):
        self.obs_history = torch.cat((self.obs_history[:, self.env.num_obs:], obs), dim=-1)

    def _pack(self, obs, privileged_obs):
        return {'obs': obs, 'privileged_obs': privileged_obs, 'obs_history': self.obs_history,
                'base_vel': self.env.get_base_vel()}

    def step(self, action):
        obs, privileged_obs, rew, done, info = self.env.step(action)
        self._roll(obs)
        return self._pack(obs, privileged_obs), rew, done, info

    def get_observations(self):
        obs = self.env.get_observations()
        privileged_obs = self.env.get_privileged_observations()
        self._roll(obs)
        return self._pack(obs, privileged_obs)

    def reset_idx(self, env_ids):
        ret = self.env.reset_idx(env_ids)
        self.obs_history[env_ids, :] = 0
        return ret

    def reset(self):
        ret = self.env.reset()
        privileged_obs = self.env.get_privileged_observations()
        self.obs_history[:, :] = 0
        return self._pack(ret, privileged_obs)

    def get_reward_buf(self):
        return self.env.get_reward_buf()
