"""HistoryWrapper: dict-observation packaging + 5-step observation history roll
(rsl_rl/rsl_rl/env/wrappers/history_wrapper.py:6-53).  No gym dependency: attribute access falls
through to the wrapped env.  On a HIP device the roll is one dtc_history_roll launch between two ping-pong
buffers (the reference returns a NEW tensor each step and callers hold the previous one until the transition is
stored, so the roll must not happen in place)."""
import torch

from ... import ops


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
        if name in ("env", "obs_history_length", "num_obs_history", "obs_history", "_spare") or "env" not in self.__dict__:
            object.__setattr__(self, name, value)
        elif hasattr(self.__dict__["env"], name) and name not in self.__dict__:
            setattr(self.__dict__["env"], name, value)      # e.g. runner sets env.episode_length_buf
        else:
            object.__setattr__(self, name, value)

    def _roll(self, obs):
        if self.obs_history.is_cuda:
            spare = self.__dict__.get("_spare")
            if spare is None or spare.shape != self.obs_history.shape:
                spare = torch.empty_like(self.obs_history)
            ops.history_roll(self.obs_history, obs.contiguous().float(), spare, self.obs_history_length)
            self._spare, self.obs_history = self.obs_history, spare
        else:
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
