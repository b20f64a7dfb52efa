"""Where the rollout side's time goes (host vs device): times act / env.step / process_env_step of the runner's loop at
4096 envs on the replay env.   python deep-tracking-control_amd/tools/rollout_profile.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd.env import ReplayEnv  # noqa: E402
from dtc_amd.runners import OnPolicyRunner  # noqa: E402

n, dev = 4096, "cuda:0"
cfg = dict(runner=dict(policy_class_name="ActorCriticDecoder", algorithm_class_name="PPO", num_steps_per_env=24, save_interval=1000),
           algorithm=dict(learning_rate=1e-3, entropy_coef=0.003), policy=dict())
r = OnPolicyRunner(ReplayEnv(n, dev), cfg, log_dir=None, device=dev)
r.learn(2)
env, alg = r.env, r.alg
obs_dict = env.get_observations()
obs, priv, hist = r._observe(obs_dict)
rew = env.get_reward_buf()
T = dict(act_host=0.0, act_total=0.0, env=0.0, store=0.0)
reps = 5
for _ in range(reps):
    for _ in range(24):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        a = alg.act(obs, priv, hist, obs_dict['base_vel'], rew)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        obs_dict, rewards, dones, infos = env.step(a)
        obs, priv, hist = r._observe(obs_dict)
        torch.cuda.synchronize(); t3 = time.perf_counter()
        alg.process_env_step(rewards, dones, next_obs=obs_dict['obs'], infos=infos)
        torch.cuda.synchronize(); t4 = time.perf_counter()
        T["act_host"] += t1 - t0; T["act_total"] += t2 - t0; T["env"] += t3 - t2; T["store"] += t4 - t3
    alg.storage.clear()
k = reps * 24
print({key: f"{v / k * 1e6:.0f} us per env step" for key, v in T.items()})
