"""The environment contract the runner consumes (rsl_rl/rsl_rl/env/vec_env.py + what HistoryWrapper and
OnPolicyRunner actually touch: history_wrapper.py:7-53, on_policy_runner.py:53-160).  Isaac Gym physics
is out of scope: any object with these members works (a real LeggedRobotDTC, or ReplayEnv below)."""
from abc import ABC, abstractmethod
from typing import Tuple, Union

import torch


class VecEnv(ABC):
    num_envs: int
    num_obs: int
    num_privileged_obs: int
    num_actions: int
    max_episode_length: int
    privileged_obs_buf: torch.Tensor
    obs_buf: torch.Tensor
    rew_buf: torch.Tensor
    reset_buf: torch.Tensor
    episode_length_buf: torch.Tensor
    extras: dict
    device: torch.device

    @abstractmethod
    def step(self, actions: torch.Tensor) -> Tuple[torch.Tensor, Union[torch.Tensor, None], torch.Tensor, torch.Tensor, dict]:
        pass

    @abstractmethod
    def reset(self, env_ids=None):
        pass

    @abstractmethod
    def get_observations(self) -> torch.Tensor:
        pass

    @abstractmethod
    def get_privileged_observations(self) -> Union[torch.Tensor, None]:
        pass


class ReplayEnv(VecEnv):
    """Stand-in for the physics side of env.step(): replays synthetic / pre-recorded observation streams
    (dtc_amd.synthetic) behind the reference env API, so the runner and PPO can be driven end to end
    without Isaac Gym.  Optionally runs the foothold planner each step exactly where
    LeggedRobotDTC.post_physics_step does (legged_robot_dtc.py:98-201) and writes foothold_obs into
    obs[:, 45:53] (legged_robot_dtc.py:270)."""

    class _Cfg:
        class env:
            num_observation_history = 5

    def __init__(self, num_envs, device, num_obs=53, num_privileged_obs=1389, num_actions=12, max_episode_length=1000,
                 seed=0, with_planner=False):
        self.cfg = ReplayEnv._Cfg()
        self.num_envs, self.num_obs, self.num_privileged_obs, self.num_actions = num_envs, num_obs, num_privileged_obs, num_actions
        self.device = device
        self.max_episode_length = max_episode_length
        self.gen = torch.Generator(device=device).manual_seed(seed)
        self.episode_length_buf = torch.zeros(num_envs, dtype=torch.long, device=device)
        self.rew_buf = torch.zeros(num_envs, device=device)
        self.reset_buf = torch.zeros(num_envs, dtype=torch.long, device=device)
        self.extras = {}
        self.with_planner = with_planner
        self._draw()

    def _draw(self):
        rn = lambda *s: torch.randn(*s, generator=self.gen, device=self.device)
        self.obs_buf = rn(self.num_envs, self.num_obs)
        self.privileged_obs_buf = rn(self.num_envs, self.num_privileged_obs).clamp_(-5, 5)
        self.base_vel = rn(self.num_envs, 3)
        if self.with_planner:
            from .. import foothold, synthetic
            sc = synthetic.scorer_inputs(self.num_envs, seed=int(torch.randint(0, 2 ** 31 - 1, (1,), generator=self.gen, device=self.device)),
                                         device=self.device)
            out = foothold.plan(sc["measured_heights"], sc["root_states"], sc["thigh_pos"], sc["commands"])
            self.foothold_obs = out["foothold_obs"]
            self.obs_buf[:, 45:53] = self.foothold_obs

    def step(self, actions):
        self.episode_length_buf += 1
        self._draw()
        self.rew_buf = 0.1 * torch.randn(self.num_envs, generator=self.gen, device=self.device)
        time_out = self.episode_length_buf > self.max_episode_length
        dones = (torch.rand(self.num_envs, generator=self.gen, device=self.device) < 0.02) | time_out
        self.reset_buf = dones.long()
        self.episode_length_buf[dones] = 0
        self.extras = {"time_outs": time_out}
        return self.obs_buf, self.privileged_obs_buf, self.rew_buf, self.reset_buf, self.extras

    def reset(self, env_ids=None):
        self.episode_length_buf.zero_()
        return self.obs_buf

    def get_observations(self):
        return self.obs_buf

    def get_privileged_observations(self):
        return self.privileged_obs_buf

    def get_base_vel(self):
        return self.base_vel

    def get_reward_buf(self):
        return self.rew_buf
