"""Soak run: OnPolicyRunner.learn on the replay env at the BASELINE size; prints throughput, memory, and checks that the
weights stay finite.   python deep-tracking-control_amd/tools/soak.py [iterations] [num_envs] [policy] [algorithm]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd.env import ReplayEnv  # noqa: E402
from dtc_amd.runners import OnPolicyRunner  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
policy = sys.argv[3] if len(sys.argv) > 3 else "ActorCriticDecoder"
algo = sys.argv[4] if len(sys.argv) > 4 else "PPO"
dev = "cuda:0"
cfg = dict(runner=dict(policy_class_name=policy, algorithm_class_name=algo, num_steps_per_env=24, save_interval=1000),
           algorithm=dict(learning_rate=1e-3, entropy_coef=0.003), policy=dict())
r = OnPolicyRunner(ReplayEnv(n, dev), cfg, log_dir=None, device=dev)
r.learn(2)
torch.cuda.synchronize()
m0 = torch.cuda.memory_allocated()
t0 = time.perf_counter()
r.learn(iters)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
sd = r.alg.actor_critic.state_dict()
ok = all(torch.isfinite(v).all().item() for v in sd.values())
print(f"{policy}/{algo}: {iters} iterations x {n} envs x 24 steps in {dt:.2f} s = {iters * n * 24 / dt:,.0f} env-steps/s "
      f"(rollout + update, replay env); lr {r.alg.learning_rate:.2e}; weights finite: {ok}; "
      f"allocated {m0 / 2**30:.2f} -> {torch.cuda.memory_allocated() / 2**30:.2f} GiB, reserved {torch.cuda.memory_reserved() / 2**30:.2f} GiB")
assert ok
