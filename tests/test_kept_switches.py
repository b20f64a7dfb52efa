"""Every environment switch the library KEEPS (switch table) runs once under its non-default value: two teacher-forced
mini-batch steps of the decoder trainer (64 envs x 24 steps) against the CPU oracle through test_hip_ppo._teacher_forced_step -- the
default path's own bounds on scalars, every parameter gradient, the weights and the learning rate.  The switches are read when the
package is imported, so every case is a process of its own ("every switch is a path nobody tests")."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import json, os, sys
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "deep-tracking-control_amd"))
import torch
from dtc_amd import synthetic as S
import test_hip_ppo as T
ref, alg = T._pair(64)
perm, e1, e2 = S.update_noise(64, 24, 4, 5, seed=123)
for k in range(2):                       # two teacher-forced mini-batches: scalars, every parameter gradient, weights, learning rate
    T._teacher_forced_step(k, ref, alg, perm[k * 384:(k + 1) * 384], e1[k], e2[k])
print("RESULT " + json.dumps(dict(ok=True, lr=alg.learning_rate)))
'''

# (all ten of them -- also DTC_OVERLAP_LANES=0, DTC_LANE_PRIO=none, DTC_LANE_POOL=1 alone -- passed in round 6; the suite keeps one
# process per distinct code path: the stream switches share one case)
CASES = [dict(DTC_H2I="0"), dict(DTC_GEMM_SPLIT="0"), dict(DTC_GEMM_SPLIT="1"), dict(DTC_H2I_CHAIN="0"),
         dict(DTC_OVERLAP_WGRAD="0", DTC_OVERLAP_LANES="0"), dict(DTC_HEADS_UNROLL="0", DTC_LANE_PRIO="none", DTC_LANE_POOL="1", DTC_ROCTX="1")]


@pytest.mark.parametrize("switch", CASES, ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_teacher_forced_step_under_a_kept_switch(switch):
    env = dict(os.environ, **switch)
    p = subprocess.run([sys.executable, "-c", f"ROOT = {ROOT!r}\n" + CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    res = json.loads([l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1][7:])
    assert res["ok"], (switch, res)
