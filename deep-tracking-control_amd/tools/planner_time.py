"""Planner timing at the bench size (98304 recorded height maps): time per call between events, fast and generic kernel.
Run it under `rocprofv3 --kernel-trace --stats` for the kernel durations."""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from dtc_amd import foothold, synthetic as S  # noqa: E402

N = int(os.environ.get("PLANNER_N", "98304"))
big = {k: v.cuda() for k, v in S.scorer_inputs(N, seed=7).items()}


def run(n=100):
    for _ in range(10):
        foothold.plan(big["measured_heights"], big["root_states"], big["thigh_pos"], big["commands"])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        foothold.plan(big["measured_heights"], big["root_states"], big["thigh_pos"], big["commands"])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"lib {os.environ.get('DTC_LIB', 'default')}: fast {run():.1f} us per call ({N} maps, host launch path included)")
if os.environ.get("PLANNER_GENERIC", "0") == "1":
    os.environ["DTC_PLANNER_GENERIC"] = "1"
    print(f"generic {run():.1f} us per call")
