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


def _bench(args, env_extra, timeout=180):
    import time
    env = dict(os.environ, **env_extra)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR"):
        env.pop(k, None)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], env=env, capture_output=True, text=True, timeout=timeout)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    return p, lines, time.monotonic() - t0


def test_more_ranks_than_gpus_fails_fast_with_one_error_line():
    """`python bench.py --gpus N` on a node with fewer than N GPUs (here: the GPUs this machine has + 1; a CPU container has
    none): ONE JSON line with "error", a non-zero exit code, well inside a minute -- no rank ever blocks in a rendezvous."""
    import torch
    n = (torch.cuda.device_count() if torch.cuda.is_available() else 0) + 1
    n = max(n, 2)
    p, lines, dt = _bench(["--gpus", str(n), "--steps", "1", "--warmup", "0"], {})
    assert p.returncode != 0 and dt < 60, (p.returncode, dt, p.stderr[-1000:])
    assert len(lines) == 1 and "error" in lines[0] and lines[0]["value"] is None and lines[0]["n_gpus"] == n, p.stdout
    assert "GPU" in lines[0]["error"] and lines[0]["metric"].startswith("env-steps/sec")


def test_a_rank_that_dies_before_the_rendezvous_produces_one_error_line():
    """Rank 1 raises before it joins; rank 0 blocks in the rendezvous until the launcher tears the job down (SIGTERM): the
    watchdog thread turns that into the error line -- the main thread never returns from the blocked call."""
    p, lines, dt = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"], dict(DTC_BENCH_LAUNCH_CHECK="1", DTC_BENCH_FAIL_RANK="1"))
    assert p.returncode != 0 and dt < 120, (p.returncode, dt, p.stderr[-1500:])
    assert len(lines) == 1 and "error" in lines[0] and lines[0]["n_gpus"] == 2, (p.stdout, p.stderr[-1500:])
    assert "simulated rank failure" in p.stderr


def test_a_rank_that_stops_taking_part_trips_the_deadline():
    """Rank 1 is alive but never issues the next collective: rank 0's phase deadline fires (10 s in this rehearsal), the line
    carries the phase it was stuck in, the job exits non-zero."""
    p, lines, dt = _bench(["--gpus", "2", "--steps", "1", "--warmup", "0"],
                          dict(DTC_BENCH_LAUNCH_CHECK="1", DTC_BENCH_HANG_RANK="1", DTC_BENCH_REHEARSAL_DEADLINE_S="10"))
    assert p.returncode != 0 and dt < 120, (p.returncode, dt, p.stderr[-1500:])
    assert len(lines) == 1 and "error" in lines[0] and "silent peer" in lines[0]["phase"], (p.stdout, p.stderr[-1500:])
