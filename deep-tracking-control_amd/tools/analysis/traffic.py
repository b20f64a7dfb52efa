"""HBM-side traffic of the GEMM family over one serialised bench step, from rocprofv3 PMC counters.

    python traffic.py collect <out_dir>     run the two counter passes (FETCH_SIZE, WRITE_SIZE; separate runs, --kernel-trace
                                            only, as MI355X_MICROARCH.md §HBM / §rocprofv3 prescribes) over
                                            `bench.py --steps 1 --warmup 0` and print the summary as JSON
    python traffic.py parse <out_dir>       re-read existing passes

bench.py imports `collect()` for its `roofline.traffic` field.

Corrections (MI355X_MICROARCH.md §HBM): FETCH_SIZE (KB) counts 64 B per 128-byte fabric request of a wide (16 B / lane)
coalesced read on gfx950 -> x2 for kernels whose loads are dwordx4 (all operand loads of the GEMM family since round 2);
WRITE_SIZE x1 (calibrated in round 1 on a launch with a known output size, profiles/r01_gemm_traffic.md).  Both counters
sit on the L2's fabric side: Infinity-Cache hits are included, so this is traffic LEAVING THE L2s, an upper bound of
the DRAM traffic.  A "step" is delimited by the foothold planner's dispatch; the LAST step of the run is the serialised
profiling step of bench.py.
"""
import csv
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
FAMILY = ("linear_fwd_kernel", "linear_dgrad_kernel", "linear_wgrad_kernel", "wgrad_group_kernel", "wgrad_reduce_kernel",
          "wgrad_group_reduce_kernel", "gru_step_fwd_kernel", "linear_s3_kernel", "wgrad_s3_group_kernel", "wgrad_s3_reduce_kernel", "wimage_kernel",
          "linear_h2i_kernel", "chain_h2i_kernel", "wgrad_h2i_group_kernel", "wgrad_h2i_reduce_kernel", "h2i_pack_kernel", "h2i_wpack_kernel",
          "gru_s3_kernel")
FETCH_FACTOR, WRITE_FACTOR = 2.0, 1.0


def _short(name):
    for f in FAMILY:
        if f in name:
            return f
    return None


def run_pass(counter, out_dir, workload="decoder", timeout=600):
    d = os.path.join(out_dir, counter)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", DTC_OVERLAP_WGRAD="0", DTC_OVERLAP_LANES="0")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", counter, "-d", d, "-o", "pmc", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-traffic",
           "--workload", workload]
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"rocprofv3 {counter} pass failed ({r.returncode}): {r.stderr.decode()[-400:]}")
    return d


def parse_pass(d, counter):
    """Sum of the counter (KB) per GEMM-family kernel over the LAST step of the run + launch counts."""
    path = None
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith("counter_collection.csv"):
                path = os.path.join(base, f)
    if path is None:
        raise RuntimeError(f"no counter_collection.csv under {d}")
    rows = []
    with open(path, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    starts = [i for i, r in enumerate(rows) if "foothold_plan" in r[1]]
    if not starts:
        raise RuntimeError("no foothold_plan dispatch in the trace")
    step = rows[starts[-1]:]
    per, n = {}, {}
    for _, name, v in step:
        k = _short(name)
        if k:
            per[k] = per.get(k, 0.0) + v
            n[k] = n.get(k, 0) + 1
    return per, n, len(step)


def summarise(out_dir):
    f, nf, _ = parse_pass(os.path.join(out_dir, "FETCH_SIZE"), "FETCH_SIZE")
    w, nw, _ = parse_pass(os.path.join(out_dir, "WRITE_SIZE"), "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(f) | set(w)):
        kernels[k] = dict(launches=nf.get(k, nw.get(k, 0)), fetch_raw_kb=round(f.get(k, 0.0), 1), write_raw_kb=round(w.get(k, 0.0), 1),
                          bytes=(f.get(k, 0.0) * FETCH_FACTOR + w.get(k, 0.0) * WRITE_FACTOR) * 1024.0)
    total = sum(v["bytes"] for v in kernels.values())
    launches = sum(v["launches"] for v in kernels.values())
    return dict(step_bytes=total, launches=launches, fetch_factor=FETCH_FACTOR, write_factor=WRITE_FACTOR, kernels=kernels,
                method="rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over the serialised last step of "
                       "bench.py --steps 1 --warmup 0; FETCH_SIZE x2 (16 B/lane loads, MI355X_MICROARCH.md §HBM), WRITE_SIZE x1 "
                       "(profiles/r01_gemm_traffic.md); fabric-side counters (Infinity-Cache hits included)")


def collect(out_dir, workload="decoder"):
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        run_pass(c, out_dir, workload)
    return _keep_summary_only(out_dir, summarise(out_dir))


def _keep_summary_only(out_dir, summary):
    """The raw rocprofv3 trees are tens of MB per pass (gpurun copies back at most 64 MiB of gpurun_out/): keep the summary as
    JSON beside them and drop the trees (DTC_KEEP_PMC=1 keeps them for `parse`)."""
    with open(os.path.join(out_dir, "summary.json"), "w") as fh:
        json.dump(summary, fh, indent=1)
    if os.environ.get("DTC_KEEP_PMC") != "1":
        for d in os.listdir(out_dir):
            if os.path.isdir(os.path.join(out_dir, d)):
                shutil.rmtree(os.path.join(out_dir, d), ignore_errors=True)
    return summary


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    res = collect(out) if mode == "collect" else summarise(out)
    print(json.dumps(res, indent=1))
