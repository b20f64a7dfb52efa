"""Round-4 probe: the foothold planner at env-step sizes (BASELINE configs[3]: 4096 envs, one launch) -- four envs per wave (rollout-buffer
default) against one env per wave (DTC_PLANNER_EPW1_MAX); HIP-event time per launch over back-to-back launches."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dtc_amd import foothold, synthetic as S  # noqa: E402

DEV = "cuda:0"
for N in (1024, 4096, 8192, 16384, 32768, 98304):
    sc = S.scorer_inputs(N, seed=7, device=DEV)
    args = (sc["measured_heights"], sc["root_states"], sc["thigh_pos"], sc["commands"])
    for _ in range(10):
        foothold.plan(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        foothold.plan(*args)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 10.0
    print(f"EPW1_MAX={os.environ.get('DTC_PLANNER_EPW1_MAX', '16384'):>6s}  N={N:6d}: {us:7.1f} us per launch  {3096.0 * N / us / 1e6:5.2f} TB/s")
