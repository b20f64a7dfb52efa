"""`python bench.py --gpus N` must launch itself (the driver's command shape has no torchrun in front): the launcher
path is rehearsed on CPU -- N ranks rendezvous over gloo on 127.0.0.1, one all-reduce, rank 0 prints the one line."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("n", [2, 4])
def test_bench_self_launches_n_ranks(n):
    env = dict(os.environ, DTC_BENCH_LAUNCH_CHECK="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "3", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout                         # ONE JSON line, from rank 0
    assert lines[0] == {"launch_check": True, "n_gpus": n, "sum": float(n), "steps": 3, "warmup": 1}


def test_bench_under_an_external_launcher_rejects_a_mismatched_world():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=2" in p.stderr
