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


def _stub_detail():
    """A full bench record of the shape bench.py builds (round 5's 21 KB line, prose and tables included)."""
    prose = "x" * 900
    return {
        "metric": "env-steps/sec, PPO.update on pre-recorded rollouts, 4096 envs, 1/2/4/8 GPU", "value": 2006737.9123456, "unit": "env-steps/s",
        "n_gpus": 8, "steps": 20, "warmup": 5, "ms_per_step": 48.98696512345, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (emulated: f16x2 split per operand, f32 accumulate)", "data": "synthetic", "gemm_arithmetic": prose,
        "gemm_accuracy": {f"k{i}": dict(fwd=1e-7, dgrad=2e-7, wgrad=3e-7) for i in range(5)},
        "gemm_accuracy_in_situ": {f"shape{i}": dict(calls=2, max_rel=1e-7, row_rel=1e-6, fp32_mfma_max_rel=1e-7) for i in range(39)},
        "config": {"workload": "BASELINE configs[1]: " + "w" * 200, "num_envs_per_gpu": 4096, "num_steps_per_env": 24, "mini_batch": 24576,
                   "epochs": 5, "parallelism": "dp8", "rccl_world": 8, "collective_sequence_ok": True,
                   "allreduce_bytes_per_step_per_rank": 378000000, "headline_at_every_n": prose,
                   "rank_ms_per_step": {"min": 48.1, "max": 49.3},
                   "world": {"backend": "nccl", "world": 8, "distinct_devices": 8, "devices": [dict(rank=i, name="AMD Instinct MI355X", pci_bus_id=f"0000:{i:02x}:00.0", uuid="u" * 36) for i in range(8)]}},
        "configs4_composite": dict(workload=prose, value=5.1e6, unit="env-steps/s", ms_per_step=123.4, steps=3, warmup=1, n_gpus=8, num_envs_total=32768, last_update=[0.1] * 7),
        "roofline": dict(bound="mfma", kernel="GEMM family: " + "k" * 300, kernel_description=prose, achieved=202.7578347026004, peak=838.8666666666667,
                         unit="TFLOP/s", frac=0.2417044838702222, peak_definition=prose, traffic=124665662.50212766, traffic_algorithmic=112042047.47,
                         launches=940, avg_launch_us=52.489377600339026, mfma_busy=0.276380276218813, measured=prose,
                         traffic_kernels={f"kern{i}": dict(launches=100, MB=1234.5) for i in range(12)},
                         mfma_busy_kernels={f"kern{i}": dict(launches=100, mfma_busy=0.3, executed_tflops=600.1, clock_ghz=2.2) for i in range(12)}),
        "roofline_planner": dict(bound="hbm", kernel="foothold_plan_fast_kernel", achieved=2864.71, peak=8000.0, unit="GB/s", frac=0.358088,
                                 traffic=301.8e6, traffic_source=prose, launches=1, avg_launch_us=106.241, bytes_per_launch=3.04e8),
        "roofline_planner_4096": dict(measured=prose, env_step_block=dict(measured=prose)),
        "kernel_classes": {f"class{i}[{i}x512x512]": dict(ms=1.234, launches=40, rate=123.4) for i in range(60)},
        "last_update": [0.0532584123, -0.00570518123, 0.0, 0.0, 0.975408123, 0.998529123, 4.3914e-05],
        "single_pass_fp32_mfma": dict(ms_per_step=93.3279, value=1053320.0, unit="env-steps/s", steps=5, note=prose),
        "cpu_baseline": dict(value=3709.75, unit="env-steps/s", cores=32, kind="port", host_cpus=256, runs_env_steps_per_s=[3700.0] * 3, sample=prose,
                             sample_short="oracle port: compute_returns + 1/5 update epochs on 4096x24, planner on 16384/98304 maps, scaled"),
    }


def test_result_line_is_short_machine_readable_and_last_on_stdout(tmp_path, capsys):
    """The driver parses the LAST stdout line and keeps a bounded tail: the line is < 4 KB whatever the full record holds (round 5's
    21 KB line left BENCH_r05.parsed null), carries every contract key, and the full record goes to the side file it names."""
    sys.path.insert(0, ROOT)
    import bench
    detail = _stub_detail()
    rel = bench.write_detail(detail, str(tmp_path / "bench_detail.json"))
    line = bench.compact_line(detail, rel)
    bench._LINE_DONE[0] = False
    assert bench.emit_line(line)
    bench._LINE_DONE[0] = False
    out = capsys.readouterr().out
    last = out.rstrip("\n").splitlines()[-1]
    assert len(last) < 4096 and "\n" not in last, len(last)
    got = json.loads(last)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline", "detail_file"):
        assert k in got, k
    assert got["value"] == pytest.approx(detail["value"], rel=1e-6) and got["ms_per_step"] == pytest.approx(detail["ms_per_step"], rel=1e-6)
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel"):
        assert k in got["roofline"], k
    assert len(got["roofline"]["kernel"]) <= 120
    assert got["roofline"]["frac"] == pytest.approx(detail["roofline"]["achieved"] / detail["roofline"]["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"], k
    assert "workload" in got["config"] and "model" not in got["config"]
    assert got["config"]["rccl_world"] == 8 and got["config"]["collective_sequence_ok"] is True
    assert got["configs4_composite"]["num_envs_total"] == 32768
    assert not any(isinstance(v, str) and len(v) > 260 for v in got.values())
    # nothing was dropped: the side file holds the whole record
    full = json.load(open(tmp_path / "bench_detail.json"))
    assert full["kernel_classes"] == detail["kernel_classes"] and len(full["gemm_accuracy_in_situ"]) == 39
