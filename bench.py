"""Headline benchmark: env-steps/sec of the PPO-update + foothold-score hot path on pre-recorded
(synthetic) rollouts, 4096 envs x 24 steps per GPU (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W            (N > 1: re-executes itself under torch.distributed.run)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one recorded rollout of 4096 envs x 24 env-steps:
    foothold planner over the 24*4096 recorded height maps  (legged_robot_dtc.py:98-201)
  + RolloutStorage.compute_returns                          (rollout_storage.py:138-152)
  + PPO.update: 5 epochs x 4 mini-batches of 24576          (ppo.py:174-357)
Inputs are resident in HBM before the timed region.  Multi-GPU = data parallel, weak scaling
(4096 envs per rank, RCCL all-reduce of the two flat gradient buckets + three scalars per step).
Rank 0 prints ONE JSON line.  N > 1: the headline stays configs[1] PER RANK at every N (the configuration BASELINE.json's
metric is quoted on: the driver's 1 -> 8 efficiency is then a like-for-like ratio), and the same line carries
`configs4_composite`: configs[4]'s model (GRU + CE-net + foothold obs) on 4096 x N envs, timed in the same job.
Failure is loud: a run that cannot complete (missing GPU, a rank that died, a hung collective, the launcher's SIGTERM) ends
with ONE JSON line carrying `"error"` and a non-zero exit code within the deadline -- never a silent hang.
"""
import argparse
import json
import os
import signal
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "deep-tracking-control_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

NUM_ENVS, NUM_STEPS = 4096, 24
FLOP_PER_ENV_STEP = 102.03e6          # 20.405 MFLOP per sample-visit x 5 epochs
PEAK_FP32_MFMA_TFLOPS = 157.3         # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, no xf32 on gfx950
PEAK_BF16_MFMA_TFLOPS = 2516.6        # MI355X_MICROARCH.md: dense bf16 (v_mfma_f32_32x32x16_bf16), 16 x the fp32 MFMA rate
PEAK_SPLIT_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0      # fp32-equivalent peak of the bf16 x 3 split path: six bf16 passes per product = 419.4
PEAK_H2_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 3.0         # ... of the two-term fp16 path (the fp16 MFMA rate = the bf16 one): three passes = 838.9


def split_peak(ops):
    return PEAK_H2_TFLOPS if ops.H2 else PEAK_SPLIT_TFLOPS


def cpu_baseline(full=False):
    """The oracle (torch-CPU restatement of the reference's PPO.update + numpy foothold planner) timed on the host cores
    as BASELINE.md §4 prescribes -- one warm-up, median of three runs -- on a BOUNDED sample of the workload: ONE of the
    update's five epochs (4 mini-batches of 24576 on the full 4096 x 24 rollout, + compute_returns) and 16384 of the
    98304 height maps; the per-env-step rate scales both parts back to the whole step (the five epochs are identical
    work).  Thread count: whichever of {min(32, cores), cores} ran the warm-up faster (torch-CPU GEMMs of this size
    stop scaling around 32 threads); `cores` reports the count actually used."""
    import statistics
    import numpy as np  # noqa: F401
    from dtc_amd import synthetic as S
    from oracle import foothold as OF
    from oracle import ppo_ref as OP
    n_cpu = os.cpu_count() or 1
    # full = True (`bench.py --cpu-baseline-full`, once per round into profiles/): the WHOLE workload -- all five epochs, all 98304
    # maps -- one warm-up + one run, so that the scaling of the sampled figure stays checked
    n_envs, n_maps, epochs = NUM_ENVS, (NUM_ENVS * NUM_STEPS if full else 16384), 5
    run_epochs = epochs if full else 1
    d = S.rollout(n_envs, NUM_STEPS, seed=4)
    perm, e1, e2 = S.update_noise(n_envs, NUM_STEPS, 4, run_epochs, seed=123)
    inp = S.scorer_inputs(n_maps, seed=7)
    args = [inp[k].numpy() for k in ("measured_heights", "root_states", "thigh_pos", "commands")]

    def one_run():
        torch.manual_seed(3)
        ac = OP.RefActorCriticDecoder()
        alg = OP.RefPPO(ac, num_learning_epochs=run_epochs, learning_rate=1e-3, entropy_coef=0.003)
        alg.init_storage(n_envs, NUM_STEPS)
        for k, v in d.items():
            if k != "last_values":
                getattr(alg.storage, k).copy_(v)
        t0 = time.perf_counter()
        alg.storage.compute_returns(d["last_values"], 0.99, 0.95)
        t_ret = time.perf_counter() - t0
        t0 = time.perf_counter()
        alg.update(perm, e1, e2)
        t_epoch = time.perf_counter() - t0
        t0 = time.perf_counter()
        for lo in range(0, n_maps, 8192):             # bounded temporaries ([chunk, 693, 4] arrays)
            OF.plan(*[x[lo:lo + 8192] for x in args], S.MEASURED_POINTS_X, S.MEASURED_POINTS_Y)
        t_sc = time.perf_counter() - t0
        per_env_step = (t_ret + (epochs // run_epochs) * t_epoch) / (n_envs * NUM_STEPS) + t_sc / n_maps
        return per_env_step, t_ret + t_epoch + t_sc

    warm = {}
    # all host cores only when that is a sane thread count for GEMMs of this size: on the 256-vCPU GPU box 256 threads ran
    # this sample at 64 env-steps/s against 3700 at 32 (measured in round 2), i.e. minutes per run
    for th in sorted({min(32, n_cpu), n_cpu if n_cpu <= 64 else min(32, n_cpu)}):
        torch.set_num_threads(th)
        warm[th] = one_run()[0]                       # doubles as the warm-up run
    cores = min(warm, key=warm.get)
    torch.set_num_threads(cores)
    runs = [one_run() for _ in range(1 if full else 3)]
    per_env_step = statistics.median(r[0] for r in runs)
    return dict(value=1.0 / per_env_step, unit="env-steps/s", cores=cores, kind="port", host_cpus=n_cpu,
                runs_env_steps_per_s=[round(1.0 / r[0], 1) for r in runs],
                sample_short=(f"oracle port: compute_returns + {run_epochs}/5 update epochs on {n_envs}x{NUM_STEPS}, planner on {n_maps}/"
                              f"{NUM_ENVS * NUM_STEPS} maps, scaled to the whole step; {runs[0][1]:.0f} s CPU/run, median of {len(runs)}"),
                sample=("WHOLE workload, nothing scaled: " if full else "") +
                       f"oracle/ppo_ref.py: compute_returns + {run_epochs} of the 5 update epochs (4 mini-batches of 24576) on {n_envs} envs x "
                       f"{NUM_STEPS} steps, oracle/foothold.py on {n_maps} of {NUM_ENVS * NUM_STEPS} height maps; "
                       f"{runs[0][1]:.1f} s of CPU work per run, warm-up + median of {len(runs)}, torch {torch.__version__} CPU, "
                       f"{cores} threads (warm-up rates: " + ", ".join(f"{k} thr {1.0 / v:.0f}" for k, v in warm.items()) + ")")


METRIC = "env-steps/sec, PPO.update on pre-recorded rollouts, 4096 envs, 1/2/4/8 GPU"
_LINE_LOCK = threading.Lock()
_LINE_DONE = [False]


def emit_line(obj) -> bool:
    """Rank 0's single JSON line (first caller wins; the watchdog thread and the main thread may race for it)."""
    with _LINE_LOCK:
        if _LINE_DONE[0]:
            return False
        _LINE_DONE[0] = True
        sys.stdout.write(json.dumps(obj, separators=(",", ":")) + "\n")
        sys.stdout.flush()
        return True


LINE_LIMIT = 4096                     # the driver keeps a bounded tail of stdout: the result line must fit well inside it
DETAIL_PATH = os.path.join(ROOT, "gpurun_out", "bench_detail.json")


def _r(x, nd=4):
    """Numbers of the result line at a sane number of significant digits (the detail file keeps full precision)."""
    if isinstance(x, float):
        return float(f"{x:.{nd + 2}g}")
    return x


def _pick(d, keys, nd=4):
    return None if d is None else {k: _r(d.get(k), nd) for k in keys if k in d}


def compact_line(detail, detail_path=None):
    """The ONE machine-readable result line (<= LINE_LIMIT bytes) out of the full record `detail`: numbers and short labels only.
    Everything else -- prose, the per-product accuracy tables, per-kernel counter tables, kernel classes -- stays in the detail
    file named by `detail_file`."""
    cfg = detail.get("config") or {}
    line = {k: _r(detail.get(k), 6) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                              "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: _r(cfg.get(k)) for k in ("workload", "num_envs_per_gpu", "num_steps_per_env", "mini_batch", "epochs", "parallelism",
                                                  "rccl_world", "collective_sequence_ok", "allreduce_bytes_per_step_per_rank") if k in cfg}
    rank_ms = cfg.get("rank_ms_per_step")
    if rank_ms and detail.get("n_gpus", 1) > 1:
        line["config"]["rank_ms_per_step"] = {k: _r(v) for k, v in rank_ms.items()}
    world = cfg.get("world")
    if world:
        line["config"]["backend"], line["config"]["distinct_devices"] = world.get("backend"), world.get("distinct_devices")
    roof = detail.get("roofline")
    if roof is not None:
        r = _pick(roof, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_algorithmic", "launches", "avg_launch_us",
                         "mfma_busy"))
        r["kernel"] = str(r.get("kernel", ""))[:120]
        line["roofline"] = r
    line["roofline_planner"] = _pick(detail.get("roofline_planner"), ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "launches",
                                                                      "avg_launch_us"))
    cpu = detail.get("cpu_baseline")
    if cpu is not None:
        line["cpu_baseline"] = dict(_pick(cpu, ("value", "unit", "cores", "kind")), sample=str(cpu.get("sample_short", cpu.get("sample", "")))[:160])
    if detail.get("single_pass_fp32_mfma") is not None:
        line["single_pass_fp32_mfma"] = _pick(detail["single_pass_fp32_mfma"], ("ms_per_step", "value"))
    c4 = detail.get("configs4_composite")
    if c4 is not None:
        line["configs4_composite"] = _pick(c4, ("value", "unit", "ms_per_step", "steps", "n_gpus", "num_envs_total"))
    if detail.get("last_update") is not None:
        line["last_update"] = [_r(x) for x in detail["last_update"]]
    line["detail_file"] = detail_path
    out = json.dumps(line, separators=(",", ":"))
    if len(out) >= LINE_LIMIT:           # cannot happen with the bounded fields above; never print an oversized line
        for k in ("last_update", "configs4_composite", "single_pass_fp32_mfma", "roofline_planner"):
            line.pop(k, None)
            out = json.dumps(line, separators=(",", ":"))
            if len(out) < LINE_LIMIT:
                break
    assert len(out) < LINE_LIMIT, len(out)
    return line


def write_detail(detail, path=None):
    """The full record (everything the result line leaves out) as a side file; returns the repo-relative path or None."""
    path = path or DETAIL_PATH
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(detail, f, indent=1)
        return os.path.relpath(path, ROOT)
    except OSError as e:
        sys.stderr.write(f"[bench] detail file not written: {e}\n")
        return None


def error_line(msg, n_gpus, phase=None, **extra):
    return dict({"metric": METRIC, "value": None, "unit": "env-steps/s", "n_gpus": n_gpus, "error": str(msg)[-1500:], "phase": phase,
                 "higher_is_better": True}, **extra)


class Watchdog(threading.Thread):
    """Turns every way an N > 1 run can stall into ONE JSON error line + a non-zero exit, from a thread that does not depend on
    the main thread returning from a blocked C call (an RCCL collective whose peer died, a communicator init, a device sync):
      * a deadline per phase (`enter(phase, seconds)`), DTC_BENCH_DEADLINE_S caps the whole run;
      * SIGTERM / SIGINT (torchrun terminates the surviving ranks when one rank fails): the C-level handler writes to a
        wake-up pipe, this thread reads it -- Python-level handlers only run once the main thread is back in the interpreter.
    Rank 0 prints the line; every rank writes a diagnostic to stderr; the process ends with os._exit."""

    def __init__(self, rank, world):
        super().__init__(daemon=True)
        self.rank, self.world = rank, world
        self.phase, self.deadline = "start", time.monotonic() + float(os.environ.get("DTC_BENCH_DEADLINE_S", "1500"))
        self.hard = self.deadline
        self.r, self.w = os.pipe()
        os.set_blocking(self.w, False)
        self.done = False

    def arm(self):
        signal.set_wakeup_fd(self.w, warn_on_full_buffer=False)
        for sig in (signal.SIGTERM, signal.SIGINT):
            signal.signal(sig, lambda *_: None)              # keep the default action from killing us before the line is out
        self.start()
        return self

    def enter(self, phase, seconds):
        self.phase, self.deadline = phase, min(self.hard, time.monotonic() + seconds)

    def finish(self):
        self.done = True

    def fail(self, msg, code=3):
        sys.stderr.write(f"[bench rank {self.rank}] {msg} (phase: {self.phase})\n")
        sys.stderr.flush()
        if self.rank == 0:
            emit_line(error_line(msg, self.world, self.phase))
        os._exit(code)

    def run(self):
        import select
        while not self.done:
            ready, _, _ = select.select([self.r], [], [], 0.5)
            if self.done:
                return
            if ready:
                sig = os.read(self.r, 16)
                self.fail(f"terminated by signal {sig[0] if sig else '?'} -- under torch.distributed.run this means ANOTHER rank "
                          "failed (its message is on stderr) and the launcher is tearing the job down", 4)
            if time.monotonic() > self.deadline:
                self.fail(f"no progress: phase '{self.phase}' exceeded its deadline -- a rank is missing or a collective hangs", 5)


def self_launch(n: int) -> int:
    """Re-execute this script as `n` ranks (one per GPU of this node) under torch.distributed.run and return its exit code.
    The children inherit stdout / stderr, so rank 0's single JSON line is what the caller of `python bench.py --gpus N`
    reads.  Rendezvous stays on 127.0.0.1 (the container host name may not resolve); the port is a free one unless
    MASTER_PORT is set."""
    import socket
    import subprocess
    launch_check = os.environ.get("DTC_BENCH_LAUNCH_CHECK") == "1"
    if not launch_check and "DTC_BENCH_DEVICE" not in os.environ:     # (DTC_BENCH_DEVICE: all ranks share one GPU, rehearsal)
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < n:                                         # fail fast, before any rank blocks in a rendezvous
            emit_line(error_line(f"--gpus {n} but this node exposes {have} GPU(s) (torch.cuda.device_count())", n, "preflight"))
            return 2
    port = os.environ.get("MASTER_PORT")
    if port is None:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's intra-node transport on this driver
    env.setdefault("OMP_NUM_THREADS", "8")                   # torchrun would pin it to 1 with a warning
    env.pop("MASTER_PORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    # rank 0's stdout is relayed line by line; if the job ends without its JSON line (a rank crashed before rank 0 could
    # report), the launcher prints the error line itself
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    seen = False
    for ln in proc.stdout:
        if ln.startswith("{"):
            seen = True
            sys.stdout.write(ln)
            sys.stdout.flush()
        else:                                                # library chatter on the ranks' stdout (gloo's connection notes) is not the result
            sys.stderr.write(ln)
    rc = proc.wait()
    if not seen:
        emit_line(error_line(f"the {n}-rank job ended with exit code {rc} and without a result line (see stderr)", n, "launcher"))
        return rc or 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the two rocprofv3 PMC passes behind roofline.traffic")
    ap.add_argument("--no-in-situ", action="store_true", help="skip the fp64 check of one step's products on their actual operands")
    ap.add_argument("--workload", default=os.environ.get("DTC_BENCH_WORKLOAD", "decoder"), choices=["decoder", "composite", "gru"],
                    help="decoder = BASELINE configs[1] (the headline, default); composite = configs[4]'s model "
                         "(GRU + CE-net + foothold obs, build-defined); gru = configs[2] (ActorCriticRecurrent, GRU 512, BPTT); "
                         "both on the same rollout shapes -- informative only.  Default from DTC_BENCH_WORKLOAD (a driver that "
                         "cannot pass flags selects configs[4]'s model with DTC_BENCH_WORKLOAD=composite)")
    ap.add_argument("--detail", default=os.environ.get("DTC_BENCH_DETAIL"), help="where the full record (everything the result line leaves "
                    "out) goes; default gpurun_out/bench_detail.json")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="time the CPU port on the WHOLE workload (no GPU work) and exit")
    a = ap.parse_args()
    if a.cpu_baseline_full:
        emit_line(dict(cpu_baseline_full=cpu_baseline(full=True)))
        return

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (the driver's command shape): become the launcher -- one rank per GPU under
        # torch.distributed.run on this node, rendezvous on 127.0.0.1; rank 0's JSON line is this process's output
        raise SystemExit(self_launch(a.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but the launcher started WORLD_SIZE={world} ranks")
    wd = Watchdog(rank, world).arm() if world > 1 else None
    try:
        run(a, rank, local_rank, world, wd)
    except SystemExit:
        raise
    except BaseException as e:                       # any failure: ONE error line from rank 0, non-zero exit, no hang in teardown
        import traceback
        traceback.print_exc()
        msg = f"rank {rank}: {type(e).__name__}: {e}"
        if wd is not None:
            wd.fail(msg, 1)
        emit_line(error_line(msg, world, "run"))
        raise SystemExit(1)
    if wd is not None:
        wd.finish()


def run(a, rank, local_rank, world, wd):
    from datetime import timedelta
    phase = (lambda name, seconds: wd.enter(name, seconds)) if wd is not None else (lambda name, seconds: None)
    coll_timeout = timedelta(seconds=int(os.environ.get("DTC_BENCH_COLL_TIMEOUT_S", "300")))
    if os.environ.get("DTC_BENCH_FAIL_RANK") == str(rank):       # test hook: this rank dies before the rendezvous
        raise RuntimeError("DTC_BENCH_FAIL_RANK: simulated rank failure")
    if os.environ.get("DTC_BENCH_LAUNCH_CHECK") == "1":
        # CPU rehearsal of the launch path only (tests/test_bench_launch.py): rendezvous, one all-reduce, rank 0's line
        phase("rendezvous (gloo rehearsal)", 120)
        dist.init_process_group("gloo", timeout=coll_timeout)
        t = torch.ones(1)
        dist.all_reduce(t)
        if os.environ.get("DTC_BENCH_HANG_RANK") == str(rank):   # test hook: this rank stops taking part
            time.sleep(3600)
        if os.environ.get("DTC_BENCH_HANG_RANK") is not None:
            phase("collective with a silent peer (rehearsal)", float(os.environ.get("DTC_BENCH_REHEARSAL_DEADLINE_S", "10")))
            dist.all_reduce(t)
        if rank == 0:
            emit_line({"launch_check": True, "n_gpus": world, "sum": float(t.item()), "steps": a.steps, "warmup": a.warmup})
        dist.destroy_process_group()
        return
    # test hooks (a 1-GPU box can rehearse the N > 1 control flow): DTC_BENCH_DEVICE pins every rank to one device,
    # DTC_BENCH_BACKEND=gloo replaces RCCL (two ranks cannot share a GPU under RCCL)
    if "DTC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["DTC_BENCH_DEVICE"])
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if local_rank >= have:
        raise RuntimeError(f"rank {rank} needs GPU {local_rank} but this node exposes {have} GPU(s)")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DTC_BENCH_BACKEND", "nccl")
        phase("rendezvous + communicator init", 240)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(dev), timeout=coll_timeout)
        else:
            dist.init_process_group(backend, timeout=coll_timeout)

    phase("library build + imports", 600)
    if int(os.environ.get("LOCAL_RANK", "0")) == 0:            # in-tree library: (re)build when missing / stale (no-op otherwise)
        import build as dtc_build
        dtc_build.build(verbose=False)
    if world > 1:
        dist.barrier()
    from dtc_amd import _ffi, foothold, h2i, ops, synthetic as S
    from dtc_amd.algorithms import PPO, RecurrentDecoderPPO, RecurrentPPO
    from dtc_amd.modules import ActorCriticDecoder, ActorCriticDecoderRecurrent, ActorCriticRecurrent

    from dtc_amd import distributed as dp
    from dtc_amd import tracing

    phase("workload set-up (weights, recorded rollout, planner inputs)", 600)
    data = S.rollout(NUM_ENVS, NUM_STEPS, seed=4 + rank, device=dev)
    # recorded planner inputs of the same rollout: one height map per (step, env)
    sc = S.scorer_inputs(NUM_ENVS * NUM_STEPS, seed=7 + rank, device=dev)
    last = {k: data[k][-1] for k in ("observations", "privileged_observations", "base_vel")}
    hid = []

    def make_workload(kind):
        """(trainer, step closure) of one workload on this rank's recorded rollout: decoder = configs[1], gru = configs[2],
        composite = configs[4]'s model."""
        torch.manual_seed(3)                      # identical initial weights on every rank
        composite, gru = kind == "composite", kind == "gru"
        if gru:
            ac = ActorCriticRecurrent(53, 1389, 12, actor_hidden_dims=[512, 256, 128], critic_hidden_dims=[512, 256, 128],
                                      activation='elu', rnn_type='gru', rnn_hidden_size=512, rnn_num_layers=1)
            alg = RecurrentPPO(ac, learning_rate=1e-3, entropy_coef=0.003, device=dev)
            alg.init_storage(NUM_ENVS, NUM_STEPS, [53], [1389], [12])
        else:
            ac = (ActorCriticDecoderRecurrent if composite else ActorCriticDecoder)(53, 1389, 12)
            alg = (RecurrentDecoderPPO if composite else PPO)(ac, learning_rate=1e-3, entropy_coef=0.003, device=dev)
            alg.init_storage(NUM_ENVS, NUM_STEPS, [53], [1389], [265], [12])
        for k, v in data.items():
            if k != "last_values" and not (gru and k == "observation_histories"):
                getattr(alg.storage, k).copy_(v)
        if (composite or gru) and not hid:               # recorded GRU states at every (step, env): the storage's saved_hidden_states
            g = torch.Generator(device=dev).manual_seed(77 + rank)
            hid.extend(0.1 * torch.randn(NUM_STEPS, 1, NUM_ENVS, 512, generator=g, device=dev) for _ in range(2))
        torch.manual_seed(123 + rank)

        def step():
            with tracing.span("foothold_plan"):
                foothold.plan(sc["measured_heights"], sc["root_states"], sc["thigh_pos"], sc["commands"])
            with tracing.span("compute_returns"):
                if gru:
                    alg.compute_returns(last["privileged_observations"])
                else:
                    alg.compute_returns(last["observations"], last["privileged_observations"], last["base_vel"])
            alg.storage.step = NUM_STEPS
            if composite or gru:
                alg.storage.saved_hidden_states_a, alg.storage.saved_hidden_states_c = [hid[0]], [hid[1]]
            with tracing.span("update"):
                return alg.update()
        return alg, step

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def prime_collectives(alg):
        """Communicator set-up and the first exchange of every bucket size happen here, never inside a timed region (also
        with --warmup 0): one all-reduce of each gradient bucket + the scalar statistics."""
        if world == 1:
            return
        arena = alg.actor_critic.ensure_arena()
        for t in ([arena.exchange_view(k) for k in arena.buckets] if hasattr(arena, "buckets") else [arena.grad_full]):
            dp.allreduce_mean_(t)                    # the real op (ReduceOp.AVG on RCCL) at the real bucket sizes
        arena.grad_full.zero_()
        dist.all_reduce(torch.zeros(1, dtype=torch.float64, device=dev))
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        """`warmup` untimed steps, then EXACTLY `steps` steps between two (barrier + device synchronise) fences; the job's time
        is the MAX over ranks.  -> (seconds, per-rank ms per step before the closing barrier, last update's return)."""
        for _ in range(warmup):
            step()
        fence()
        if ops.AMAX_STATS is not None:
            ops.AMAX_STATS.clear()
        t0 = time.perf_counter()
        for _ in range(steps):
            out = step()
        torch.cuda.synchronize()
        mine = time.perf_counter() - t0                  # this rank's own time (before the closing barrier)
        if ops.AMAX_STATS:                               # DTC_AMAX_STATS=1 (debug): operands of the fp16 path that no kernel published an amax for
            for k, v in sorted(ops.AMAX_STATS.items(), key=lambda kv: -kv[1]):
                sys.stderr.write(f"amax fallback {k}: {v / steps:.1f} per step\n")
            sys.stderr.write(f"amax fallbacks per step: {sum(ops.AMAX_STATS.values()) / steps:.1f}\n")
            ops.AMAX_STATS.clear()
        fence()
        elapsed = time.perf_counter() - t0
        rank_ms = [mine / steps * 1e3]
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
            tl = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
            dist.all_gather(tl, torch.tensor([mine / steps * 1e3], dtype=torch.float64, device=dev))
            rank_ms = [float(x.item()) for x in tl]
        return elapsed, rank_ms, out

    composite, gru = a.workload == "composite", a.workload == "gru"
    alg, step = make_workload(a.workload)
    phase("first exchange of every gradient bucket", 240)
    prime_collectives(alg)
    phase(f"warm-up + timed region ({a.warmup} + {a.steps} steps)", 120 + 3.0 * (a.warmup + a.steps))
    elapsed, rank_ms, out = timed(step, a.steps, a.warmup)
    phase("accuracy check, per-kernel pass", 900)

    # per-kernel HIP-event timing of one more step (same process, each kernel on the stream it is launched on)
    # for the roofline object.  The timed region above overlaps the weight-gradient GEMMs with the
    # data-gradient chain on a second stream; a kernel that shares the chip has no duration of its own, so this
    # pass runs the same kernels serialised (overlap off) -- rocprofv3 cross-check: DTC_OVERLAP_WGRAD=0.
    roof, classes, planner, planner_4096 = None, None, None, None
    accuracy = None
    if rank == 0:
        # accuracy of both GEMM paths against fp64 on the bench's 512-wide layer (outside the timed region; the fp64 product is
        # the CHECKER, computed by torch): max |y - y64| / max |y64| for the forward, data-gradient and weight-gradient products
        g = torch.Generator(device=dev).manual_seed(5)
        Mx, Nx, Kx = 24576, 512, 512
        Xa = torch.randn(Mx, Kx, device=dev, generator=g)
        Wa = torch.randn(Nx, Kx, device=dev, generator=g) / Kx ** 0.5
        dZa = torch.randn(Mx, Nx, device=dev, generator=g) / Mx ** 0.5
        refs = dict(fwd=Xa.double() @ Wa.double().T, dgrad=dZa.double() @ Wa.double(), wgrad=dZa.double().T @ Xa.double())
        accuracy = {}
        h2_was = ops.H2
        for name, sp, h2 in (("fp32_mfma", False, h2_was), ("split_bf16x3", True, False), ("split_f16x2", True, True)):
            ops.set_split(ops.SPLIT, h2=h2)
            if sp and ops.H2 != h2:                  # (DTC_S3_WIMG=0: no fp16 kernels)
                continue
            Y, dX, dW, db = (torch.empty(Mx, Nx, device=dev), torch.empty(Mx, Kx, device=dev), torch.empty(Nx, Kx, device=dev),
                             torch.empty(Nx, device=dev))
            ops.linear_fwd(Xa, Wa, None, Y, None, split=sp)
            ops.linear_dgrad(dZa, Wa, dX, split=sp)
            jobs = [(dZa, Xa, dW, db)]
            ws = ops.workspace(ops.wgrad_group_workspace_bytes(jobs, Mx, split=sp), dev)
            ops.wgrad_group(jobs, Mx, ws, split=sp)
            err = lambda y, r: float((y.double() - r).abs().max() / r.abs().max())
            accuracy[name] = dict(fwd=err(Y, refs["fwd"]), dgrad=err(dX, refs["dgrad"]), wgrad=err(dW, refs["wgrad"]))
        ops.set_split(ops.SPLIT, h2=h2_was)
        if ops.SPLIT:
            Xi, dZi = h2i.HImage.from_tensor(Xa), h2i.HImage.from_tensor(dZa)
            Y, dX, dW, db = (torch.empty(Mx, Nx, device=dev), torch.empty(Mx, Kx, device=dev), torch.empty(Nx, Kx, device=dev),
                             torch.empty(Nx, device=dev))
            h2i.linear_fwd(Xi, Wa, None, Y, None, None)
            h2i.linear_dgrad(dZi, Wa, dX)
            jobs = [(dZi, Xi, dW, 0, db)]
            h2i.wgrad_group(jobs, Mx, ops.workspace(h2i.wgrad_group_workspace_bytes(jobs, Mx), dev))
            err = lambda y, r: float((y.double() - r).abs().max() / r.abs().max())
            accuracy["operand_images_f16x2"] = dict(fwd=err(Y, refs["fwd"]), dgrad=err(dX, refs["dgrad"]), wgrad=err(dW, refs["wgrad"]))
        accuracy["measure"] = ("max |y - y_fp64| / max |y_fp64| on a 24576 x 512 x 512 layer, random normal operands (the friendliest "
                               "distribution: see gemm_accuracy_in_situ for the step's real operands)")
        del Xa, Wa, dZa, refs
    if world == 1 and rank == 0:
        # BASELINE configs[3]: ONE planner launch over 4096 envs x 4 legs (12.7 MB: launch / latency-bound, not HBM-bound):
        # its own line, timed with events on the launch stream over 50 back-to-back launches
        sc4 = {k: v[:NUM_ENVS].contiguous() for k, v in sc.items()}
        for _ in range(5):
            foothold.plan(sc4["measured_heights"], sc4["root_states"], sc4["thigh_pos"], sc4["commands"])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            foothold.plan(sc4["measured_heights"], sc4["root_states"], sc4["thigh_pos"], sc4["commands"])
        e1.record()
        torch.cuda.synchronize()
        call_us = e0.elapsed_time(e1) * 1e3 / reps
        # the kernel itself: HIP events of the in-library profiler around each launch on its stream (the back-to-back figure above is
        # the launch path of a call -- ctypes marshalling + dispatch -- at this size: one env per wave instead of four changes neither)
        _ffi.lib().dtc_prof_reset()
        _ffi.lib().dtc_prof_enable(1)
        for _ in range(reps):
            foothold.plan(sc4["measured_heights"], sc4["root_states"], sc4["thigh_pos"], sc4["commands"])
        torch.cuda.synchronize()
        _ffi.lib().dtc_prof_enable(0)
        pl4 = [r for r in _ffi.prof_report() if r["name"].split("[")[0] == "foothold_plan"]
        _ffi.lib().dtc_prof_reset()
        us = sum(r["ms_total"] for r in pl4) * 1e3 / max(1, sum(r["launches"] for r in pl4)) if pl4 else call_us
        planner_4096 = dict(bound="hbm", kernel="foothold_plan_fast_kernel", workload="BASELINE configs[3]: 4096 envs x 4 legs, one launch",
                            bytes_per_launch=3096.0 * NUM_ENVS, avg_launch_us=us, achieved=3096.0 * NUM_ENVS / (us * 1e-6) / 1e9,
                            peak=8000.0, unit="GB/s", frac=3096.0 * NUM_ENVS / (us * 1e-6) / 1e9 / 8000.0,
                            env_steps_per_s=NUM_ENVS / (call_us * 1e-6), call_us=call_us,
                            measured="avg_launch_us: HIP events around each of 50 launches on the launch stream (incl. ~3 us of event overhead); "
                                     "call_us / env_steps_per_s: 50 back-to-back calls through the Python / ctypes boundary (launch-path bound)")
        # the whole post-physics block of one env step at this size: ONE launch (foothold.EnvStep: planner + check_termination +
        # foothold rewards + compute_observations on the env's buffers) against the four separate launches
        es = S.env_state(NUM_ENVS, seed=13, device=dev)
        gq = torch.Generator(device=dev).manual_seed(90)
        es["foot_positions"] = torch.cat([es["root_states"][:, None, :2] + 0.3 * torch.randn(NUM_ENVS, 4, 2, generator=gq, device=dev),
                                          0.05 * torch.randn(NUM_ENVS, 4, 1, generator=gq, device=dev)], dim=2)
        es["contact_filt"] = torch.rand(NUM_ENVS, 4, generator=gq, device=dev) < 0.6
        fused = foothold.EnvStep(NUM_ENVS, dev)
        fkw = {k: es[k] for k in ("root_states", "thigh_pos", "commands", "contact_forces", "termination_contact_indices", "episode_length_buf",
                                  "projected_gravity", "foot_positions", "contact_filt", "base_ang_vel", "dof_pos", "default_dof_pos", "dof_vel",
                                  "actions", "forces", "height_noise_offset", "u_obs", "noise_scale_vec", "u_heights", "measured_heights")}

        def four_launches():
            p = foothold.plan(es["measured_heights"], es["root_states"], es["thigh_pos"], es["commands"])
            foothold.check_termination(es["contact_forces"], es["termination_contact_indices"], es["episode_length_buf"], 1000,
                                       es["projected_gravity"], es["root_states"], es["measured_heights"])
            foothold.rewards(es["foot_positions"], p["optimal_footholds_world"], es["contact_filt"])
            foothold.compute_observations(es["base_ang_vel"], es["projected_gravity"], es["commands"], es["dof_pos"], es["default_dof_pos"],
                                          es["dof_vel"], es["actions"], p["foothold_obs"], es["root_states"], es["measured_heights"],
                                          es["forces"], es["height_noise_offset"], es["u_obs"], es["noise_scale_vec"], es["u_heights"])

        def per_call_us(fn):
            for _ in range(5):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / reps * 1e6
        us_fused, us_four = per_call_us(lambda: fused(max_episode_length=1000, **fkw)), per_call_us(four_launches)
        planner_4096["env_step_block"] = dict(
            workload="one env step's post-physics block at 4096 envs: planner + check_termination + foothold rewards + compute_observations",
            one_launch_us=us_fused, four_launches_us=us_four, env_steps_per_s=NUM_ENVS / (us_fused * 1e-6),
            measured="wall clock of 50 back-to-back calls incl. the Python / ctypes path, device synchronised at both ends; same outputs bit "
                     "for bit (tests/test_hip_envstep.py)")

    # EVERY rank runs these two extra steps (they contain the data-parallel collectives); only rank 0 records events
    lib = _ffi.lib()
    overlap, overlap_rec = getattr(alg, "overlap_wgrad", False), getattr(alg, "overlap", False)
    alg.overlap_wgrad = alg.overlap = False
    in_situ = None
    if rank == 0 and world == 1 and a.workload == "decoder" and getattr(alg, "use_images", False) and not a.no_in_situ:
        # every wide product of one serialised step checked against fp64 ON ITS ACTUAL OPERANDS (forward: the layer's real input image,
        # weights and bias; data / weight gradients: the step's real, heavy-tailed dZ), right after its launch, next to the single-pass
        # fp32 MFMA kernel on the same operands.  First two calls of every (kind, shape); outside the timed region.
        h2i.capture_begin(per_key=2)
        try:
            step()
            torch.cuda.synchronize()
        finally:
            rows = h2i.capture_end()
        worst = lambda key, which, f: max((r[which][f] for r in rows[key] if r[which] is not None), default=None)
        in_situ = {k: dict(calls=len(v), max_rel=worst(k, "h2i", "max_rel"), row_rel=worst(k, "h2i", "row_rel"),
                           fp32_mfma_max_rel=worst(k, "fp32_mfma", "max_rel"), fp32_mfma_row_rel=worst(k, "fp32_mfma", "row_rel"),
                           rows_span_decades=max(r["ref_rows_span"] for r in v), zero_rows=max(r["zero_rows"] for r in v))
                   for k, v in sorted(rows.items())}
        in_situ["measure"] = ("per product of one serialised bench step (first two calls of each shape): max_rel = max |y - y64| / max |y64|, "
                              "row_rel = max over rows of (max |y - y64| of the row / max |y64| of the row), y64 = fp64 product of the SAME "
                              "operands (decoded operand images, the step's real X / dZ / W / b; activation, sign record and added terms "
                              "included); fp32_mfma_* = the single-pass v_mfma_f32_32x32x2_f32 kernels on those operands; rows_span_decades = "
                              "log10(largest / smallest non-zero row magnitude of the result), zero_rows = rows of the result that are exactly zero")
    dp.trace_collectives(True)                       # N > 1: what one step exchanges, and that all ranks issue the same sequence
    step()
    torch.cuda.synchronize()
    coll_log = dp.assert_same_collective_sequence()      # raises on any rank whose sequence differs
    dp.trace_collectives(False)
    world_facts = None
    if world > 1:
        # what the job actually ran on: the communicator's backend and size, and every rank's device (PCI bus id + name) --
        # two ranks on one device, or a gloo rehearsal, must be visible in the line
        pr = torch.cuda.get_device_properties(local_rank)
        mine_dev = dict(rank=rank, local_rank=local_rank, name=pr.name,
                        pci_bus_id="%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", 0), getattr(pr, "pci_device_id", 0)),
                        uuid=str(getattr(pr, "uuid", "")))
        devs = [None] * world
        dist.all_gather_object(devs, mine_dev)
        world_facts = dict(backend=dist.get_backend(), world=dist.get_world_size(), devices=devs,
                           distinct_devices=len({d["pci_bus_id"] for d in devs}))
    if rank == 0:
        lib.dtc_prof_reset()
        lib.dtc_prof_enable(1)
    step()
    torch.cuda.synchronize()
    lib.dtc_prof_enable(0)
    alg.overlap_wgrad, alg.overlap = overlap, overlap_rec
    if world > 1:
        dist.barrier()
    if rank == 0:
        rep = _ffi.prof_report()
        lib.dtc_prof_reset()
        # the GEMM family = forward / data-gradient / weight-gradient kernels AND the split-reduce kernels the weight
        # gradients need (their time counts against the family's FLOP; they add no FLOP of their own)
        fam = ("linear_fwd", "linear_dgrad", "linear_fwd_chain", "linear_dgrad_chain", "linear_wgrad", "wgrad_reduce", "gru_step_fwd", "wimage",
               "h2i_pack")
        gemm = [r for r in rep if r["name"].split("[")[0] in fam]
        # (the composite's GRU steps call the same three kernels from inside dtc_gru_fwd / dtc_gru_bwd)
        ms = sum(r["ms_total"] for r in gemm)
        fl = sum(r["work"] for r in gemm if not r["name"].startswith(("wgrad_reduce", "wimage", "h2i_pack")))
        n_launch = sum(r["launches"] for r in gemm)
        algo_bytes = sum(r["bytes"] for r in gemm)
        achieved = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        split = ops.SPLIT
        peak = split_peak(ops) if split else PEAK_FP32_MFMA_TFLOPS
        if split and getattr(alg, "use_images", False):
            kernel_desc = ("linear_h2i_kernel<FWD | DGRAD | MSE> (both operands as block-scaled fp16 (hi, lo) images by LDS-DMA, no conversion in "
                           "any K loop; 64-row tiles for launches of at most one 128 x 128 tile per CU), wgrad_h2i_group_kernel (+ its reduce "
                           "kernel), h2i_wpack_kernel (weight images, one launch per phase), h2i_pack_kernel (rollout rows and fp32 hand-over "
                           "tensors -> images); csrc/gemm_h2i.hip, wgrad_h2i.hip: three v_mfma_f32_32x32x16_f16 passes per product, fp32 "
                           "accumulate; the narrow layers (< 128 columns) run through the same kernels and are part of the family"
                           + ("; the GRU time steps (gru_s3_kernel: three bf16 terms / six passes, csrc/gru_s3.hip) are part of the family too"
                              if (composite or gru) else ""))
        elif split and ops.H2:
            kernel_desc = ("linear_s3_kernel<.., H2> (+ the weight-image launches), wgrad_s3_group_kernel<.., H2> (+ its reduce kernel): "
                           "split-precision GEMM family -- every fp32 operand scaled by a power of two from its tensor's amax and written as "
                           "two fp16 terms, three v_mfma_f32_32x32x16_f16 passes per product, fp32 accumulate (csrc/s3_core.hpp, gemm_s3.hip, "
                           "wgrad_s3.hip); the recurrent kernels (gru_s3_kernel) keep three bf16 terms / six passes; the narrow layers "
                           "(< 128 columns) stay on the single-pass v_mfma_f32_32x32x2_f32 kernels and are part of the family")
        elif split:
            kernel_desc = ("linear_s3_kernel (+ the weight-image launches), wgrad_s3_group_kernel (+ its reduce kernel), gru_s3_kernel: "
                           "split-precision GEMM family -- every fp32 operand as three bf16 terms, six v_mfma_f32_32x32x16_bf16 passes per "
                           "product, fp32 accumulate (csrc/gemm_s3.hip, wgrad_s3.hip, gru_s3.hip); the narrow layers (< 128 columns) stay on "
                           "the single-pass v_mfma_f32_32x32x2_f32 kernels and are part of the family")
        else:
            kernel_desc = "linear_{fwd,dgrad}_kernel, wgrad_group_kernel (+ its split-reduce kernel): fp32 v_mfma_f32_32x32x2_f32 GEMM family"
        kernel_short = ("GEMM family: linear_h2i_kernel + wgrad_h2i_group_kernel (+reduce, image packs), 3 x v_mfma_f32_32x32x16_f16"
                        if split and getattr(alg, "use_images", False) else
                        "GEMM family: linear_s3_kernel + wgrad_s3_group_kernel, split-precision MFMA" if split else
                        "GEMM family: linear_{fwd,dgrad}_kernel + wgrad_group_kernel, v_mfma_f32_32x32x2_f32")
        roof = dict(bound="mfma", kernel=kernel_short, kernel_description=kernel_desc,
                    achieved=achieved, peak=peak, unit="TFLOP/s", frac=achieved / peak,
                    peak_definition=(("dense fp16 / bf16 MFMA peak 2516.6 TFLOP/s / 3 passes = 838.9 TFLOP/s of fp32-equivalent work "
                                      "(MI355X_MICROARCH.md); achieved counts the ALGORITHMIC fp32 FLOP (2 M N K per product), not the fp16 "
                                      "passes.  (Rounds 2-4 priced the family against 419.4 = the six-pass bf16 x 3 roof: DTC_GEMM_SPLIT=1.)")
                                     if ops.H2 else
                                     ("dense bf16 MFMA peak 2516.6 TFLOP/s / 6 passes = 419.4 TFLOP/s of fp32-equivalent work "
                                      "(MI355X_MICROARCH.md); achieved counts the ALGORITHMIC fp32 FLOP (2 M N K per product), not the bf16 passes"))
                    if split else "fp32 MFMA peak (MI355X_MICROARCH.md)",
                    frac_of_bf16x3_roof=achieved / PEAK_SPLIT_TFLOPS,
                    frac_of_fp32_mfma_peak=achieved / PEAK_FP32_MFMA_TFLOPS,
                    traffic=None, traffic_algorithmic=algo_bytes / max(1, n_launch), launches=n_launch,
                    measured="HIP events per launch, kernels serialised on one stream; the split-reduce launches of the weight "
                             "gradients and the weight-image launches are part of the family (time, no FLOP)",
                    avg_launch_us=ms * 1e3 / max(1, n_launch), flop_per_launch=fl / max(1, n_launch),
                    reduce_ms=sum(r["ms_total"] for r in gemm if r["name"].startswith("wgrad_reduce")))
        if world == 1 and ops.SPLIT:
            # the rate the matrix pipe SUSTAINS for this family's MFMA stream (registers only, nothing else in the loop): the chip clocks
            # to its power budget, so the all-zero run shows the instruction-stream ceiling and the random-operand run the ceiling on
            # real data -- both far from the data sheet's dense-bf16 peak the `peak` / `frac` fields are priced against
            try:
                z, r = ops.mfma_sustained(dev, False), ops.mfma_sustained(dev, True)
                roof["sustained_mfma"] = dict(zero_operands=z, random_operands=r,
                                              unit="TFLOP/s of fp32-equivalent work (fp16 FLOP / 3)" if ops.H2 else "TFLOP/s of fp32-equivalent work (bf16 FLOP / 6)",
                                              measured=("dtc_probe_mfma_stream_h2: 768 workgroups x 4 waves, 12 v_mfma_f32_32x32x16_f16 per stage on "
                                                        if ops.H2 else
                                                        "dtc_probe_mfma_stream: 768 workgroups x 4 waves, 24 v_mfma_f32_32x32x16_bf16 per stage on ") +
                                                       "register operands, 12 launches of ~2-3 ms after 3 warm-up launches, HIP events")
                roof["frac_of_sustained_random"] = achieved / r if r > 0 else None
            except Exception as e:
                roof["sustained_mfma_error"] = str(e)[-200:]
        if world == 1 and not a.no_traffic:
            # HBM-side traffic of the same family over one serialised step: two rocprofv3 PMC passes over a child bench run
            try:
                sys.path.insert(0, os.path.join(ROOT, "deep-tracking-control_amd", "tools", "analysis"))
                import traffic as TR
                tr = TR.collect(os.path.join(ROOT, "gpurun_out", "traffic_pmc"), a.workload)
                roof["traffic"] = tr["step_bytes"] / max(1, tr["launches"])
                roof["traffic_step_bytes"], roof["traffic_algorithmic_step_bytes"] = tr["step_bytes"], algo_bytes
                roof["traffic_over_algorithmic"] = tr["step_bytes"] / algo_bytes if algo_bytes > 0 else None
                roof["traffic_launches"] = tr["launches"]
                roof["traffic_source"] = tr["method"]
                roof["traffic_kernels"] = {k: dict(launches=v["launches"], MB=round(v["bytes"] / 1e6, 1)) for k, v in tr["kernels"].items()}
            except Exception as e:                 # no rocprofv3 on this box / counters unavailable: say so, keep the line
                roof["traffic_error"] = str(e)[-300:]
            # MFMA-pipe busy fraction of the same family over one serialised step: two more PMC passes (SQ / GRBM counters)
            try:
                import gemm_pmc as GP
                pm = GP.collect(os.path.join(ROOT, "gpurun_out", "gemm_pmc"), a.workload)
                roof["mfma_busy"] = pm["family_mfma_busy"]
                roof["mfma_busy_source"] = ("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE ... over the serialised "
                                            "last step of a child bench run: busy cycles / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), "
                                            "summed over the family's launches (profiled runs clock lower than the timed region)")
                roof["mfma_busy_kernels"] = {k: dict(launches=v["launches"], mfma_busy=round(v["mfma_busy"], 3),
                                                     executed_tflops=round(v["executed_tflops"], 1), clock_ghz=round(v["clock_ghz"], 2))
                                             for k, v in pm["kernels"].items() if v["ms"] > 0.5}
            except Exception as e:
                roof["mfma_busy_error"] = str(e)[-300:]
        pl = [r for r in rep if r["name"].split("[")[0] == "foothold_plan"]
        if pl:                                  # the HBM-side kernel of the path: bytes = 3096 B/env
            pms, pby, pn = (sum(r[k] for r in pl) for k in ("ms_total", "work", "launches"))
            planner = dict(bound="hbm", kernel="foothold_plan_fast_kernel", achieved=pby / (pms * 1e-3) / 1e9, peak=8000.0,
                           unit="GB/s", frac=pby / (pms * 1e-3) / 1e9 / 8000.0,
                           traffic=301.8e6 * (pby / pn) / (3096.0 * 98304),
                           traffic_source="rocprofv3 PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/r05_planner_pmc.md (round 5 re-collection, "
                                          "tools/jobs/r5_refresh_profiles.sh; the kernel is unchanged since round 2) "
                                          "(instruction-bound: SQ_ACTIVE_INST_VALU = 86 % of the kernel's cycles; rocprofv3 "
                                          "kernel durations over 30 launches: 104.9 us average, 102.4 us minimum)",
                           launches=pn, avg_launch_us=pms * 1e3 / pn, bytes_per_launch=pby / pn)
        classes = {r["name"]: dict(ms=round(r["ms_total"], 3), launches=r["launches"],
                                   rate=(r["work"] / (r["ms_total"] * 1e-3) / 1e12) if r["ms_total"] > 0 else 0.0)
                   for r in rep}

    # N > 1, headline workload: configs[4]'s model (GRU + CE-net + foothold obs) on the same ranks in the same job -- 4096 envs
    # per rank, i.e. 32768 envs at N = 8 -- so that the first multi-GPU run measures BOTH the like-for-like weak-scaling
    # headline and the configuration BASELINE.json names for 8 GPUs.  Every rank takes part (its update exchanges buckets).
    configs4 = None
    if world > 1 and a.workload == "decoder" and os.environ.get("DTC_BENCH_CONFIGS4", "1") != "0":
        phase("configs[4] composite: set-up + first exchanges", 600)
        del alg, step
        torch.cuda.empty_cache()
        alg4, step4 = make_workload("composite")
        prime_collectives(alg4)
        k4 = max(1, min(a.steps, 3))
        phase(f"configs[4] composite: warm-up + timed region (1 + {k4} steps)", 180 + 6.0 * k4)
        el4, rank_ms4, out4 = timed(step4, k4, 1)
        configs4 = dict(workload=f"BASELINE configs[4]: {NUM_ENVS * world} envs data-parallel over {world} GPUs ({NUM_ENVS} per rank) x "
                                 f"{NUM_STEPS} steps, GRU + CE-net + foothold-obs composite (build-defined, DESIGN.md): planner + "
                                 "compute_returns + RecurrentDecoderPPO.update (5 epochs x 4 recurrent mini-batches, BPTT), gradient "
                                 "buckets all-reduced over RCCL",
                        value=NUM_ENVS * NUM_STEPS * world * k4 / el4, unit="env-steps/s", ms_per_step=el4 / k4 * 1e3, steps=k4, warmup=1,
                        n_gpus=world, num_envs_total=NUM_ENVS * world,
                        rank_ms_per_step={"min": min(rank_ms4), "max": max(rank_ms4)}, last_update=[float(x) for x in out4])
        alg = alg4
    phase("result line", 120)
    if rank == 0:
        env_steps = NUM_ENVS * NUM_STEPS * world * a.steps
        value = env_steps / elapsed
        line = {
            "metric": METRIC,
            "value": value, "unit": "env-steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": elapsed / a.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("f32 (emulated: f16x2 split per operand, f32 accumulate)" if (ops.SPLIT and (ops.H2 or getattr(alg, "use_images", False)))
                      else "f32 (emulated: bf16x3 split per operand, f32 accumulate)" if ops.SPLIT else "f32"),
            "data": "synthetic",
            "gemm_arithmetic": ("results, accumulation, narrow layers, losses and optimiser in f32; the operands of the wide layers (>= 128 columns) live "
                                "in HBM as block-scaled two-term fp16 images (x 2^e = hi + lo: 22 significant bits; e per ROW and block of 128 "
                                "columns from that block's own largest finite element, weights per 128 x 128 block), 3 fp16 MFMA passes per "
                                "product (exact in the f32 accumulator; dropped lo lo' = 2^-22 of a product), accumulator rows rescaled exactly "
                                "at block borders: every row is as accurate relative to ITSELF as an f32 dot product, whatever the other rows "
                                "hold, and a non-finite element poisons only its own row / column (csrc/h2i_core.hpp).  Measured against fp64: "
                                "gemm_accuracy (random operands) and gemm_accuracy_in_situ (the step's real operands).  DTC_H2I=0: round 4's "
                                "converting kernels (tensor-amax scale); DTC_GEMM_SPLIT=0: single-pass fp32 MFMA everywhere"
                                if getattr(alg, "use_images", False) and ops.SPLIT else
                                ("f32 operands and results; wide layers on the fp16 matrix pipe as 2-term splits (x 2^e = hi + lo, e from the "
                                 "tensor's amax: 22 significant bits, the sums scaled back exactly), 3 MFMA passes, fp32 accumulate -- the error "
                                 "level of the fp32 MFMA chain, measured below against fp64 next to the single-pass fp32 MFMA kernels and the "
                                 "bf16 x 3 / six-pass kernels (DTC_GEMM_SPLIT=1); DTC_GEMM_SPLIT=0 selects the single-pass kernels everywhere")
                                if ops.H2 else
                                ("f32 operands and results; wide layers on the bf16 matrix pipe as 3-term splits (a = a1 + a2 + a3 exactly to 2^-24), "
                                 "6 MFMA passes, fp32 accumulate -- fp32-level accuracy, measured below against fp64 next to the single-pass "
                                 "fp32 MFMA kernels; DTC_GEMM_SPLIT=0 selects the single-pass kernels everywhere") if ops.SPLIT else
                                "f32 single-pass v_mfma_f32_32x32x2_f32 (DTC_GEMM_SPLIT=0)"),
            "gemm_accuracy": accuracy,
            "gemm_accuracy_in_situ": in_situ,
            "config": {"workload": ("BASELINE configs[1]: 4096 envs x 24 steps/GPU, ActorCriticDecoder (CE-net + terrain encoder 512 + MLP "
                                    "heads): planner over 98304 recorded maps + compute_returns + PPO.update (5 epochs x 4 x 24576)")
                       if not (composite or gru) else
                       ("BASELINE configs[2]: ActorCriticRecurrent (GRU 512), 4096 envs x 24 steps: planner + compute_returns + "
                        "RecurrentPPO.update (5 epochs x 4 recurrent mini-batches of 1024 envs x 24 steps, BPTT)") if gru else
                       ("BASELINE configs[4] model (GRU + CE-net + foothold obs), 4096 envs x 24 steps/GPU: planner + compute_returns + "
                        "RecurrentDecoderPPO.update (5 epochs x 4 recurrent mini-batches of 1024 envs x 24 steps, BPTT)"),
                       "num_envs_per_gpu": NUM_ENVS, "num_steps_per_env": NUM_STEPS, "mini_batch": 24576,
                       "epochs": 5, "wgrad_overlap_stream": bool(overlap),
                       "headline_at_every_n": "configs[1] per rank (weak scaling: the N = 1 workload on every GPU); configs[4]'s model on "
                                              "4096 x N envs is measured in the same job at N > 1: configs4_composite", "parallelism": f"dp{world}" if world > 1 else "single",
                       "mfma_frac_whole_step": None if (composite or gru) else
                       (FLOP_PER_ENV_STEP * value / world) / ((split_peak(ops) if ops.SPLIT else PEAK_FP32_MFMA_TFLOPS) * 1e12),
                       "rank_ms_per_step": {"min": min(rank_ms), "max": max(rank_ms)},
                       "allreduce_bytes_per_step_per_rank": dp.bytes_reduced(coll_log) if world > 1 else 0,
                       "collectives_per_step": len(coll_log),
                       "collective_sequence_ok": True,      # assert_same_collective_sequence() above passed on every rank
                       "rccl_world": (world_facts["world"] if world_facts and world_facts["backend"] == "nccl" else None),
                       "world": world_facts},
            "configs4_composite": configs4,
            "roofline": roof,
            "roofline_planner": planner,
            "roofline_planner_4096": planner_4096,
            "kernel_classes": classes,
            "last_update": [float(x) for x in out],
        }
        if world == 1 and ops.SPLIT and not a.no_traffic:
            # the SAME workload on the single-pass fp32 MFMA kernels, timed in this process after everything above (same box, same
            # data): for a reader who wants the headline without the split arithmetic.  (Not in the --no-traffic child runs of the
            # counter passes: their parsers take the LAST step of the process, which must stay the serialised split-path step.)
            ops.set_split(False)
            try:
                for _ in range(2):
                    step()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                k = max(3, min(a.steps, 5))
                for _ in range(k):
                    step()
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t0) / k
                line["single_pass_fp32_mfma"] = dict(ms_per_step=dt * 1e3, value=NUM_ENVS * NUM_STEPS / dt, unit="env-steps/s", steps=k,
                                                     note="DTC_GEMM_SPLIT=0 path (v_mfma_f32_32x32x2_f32 everywhere), same process and data")
            finally:
                ops.set_split(True)
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        emit_line(compact_line(line, write_detail(line, a.detail)))
    if world > 1:
        phase("teardown", 60)
        dist.barrier()
        if wd is not None:
            wd.finish()                              # the line is out: a slow communicator teardown is not a failure
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
