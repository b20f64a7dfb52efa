"""gpurun_out/r2p (written by tools/jobs/r2_refresh_profiles.sh) -> profiles/r02_*  (run from the repo root).

    python deep-tracking-control_amd/tools/analysis/collect_profiles.py [src_dir] [tag]

  r02_bench_n1.json                     the default bench line, pretty-printed
  r02_bench_{gru,composite}_informative.json
  r02_gemm_shapes.md                    per-shape table of the serialised profiling pass (DTC_PROF_SHAPES=1)
  r02_gemm_traffic_step.md              roofline.traffic_* of the bench line as a table
  r02_bench_kernel_stats_{serial,overlap}.csv   rocprofv3 --stats summaries with shortened kernel names
  r02_dp_rehearsal.log, r02_soak.log    copied
"""
import csv
import glob
import json
import os
import re
import shutil
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/r3p"
tag = sys.argv[2] if len(sys.argv) > 2 else "r03"
rnd = int(tag[1:])
dst = "profiles"


def last_json(path):
    """The result line of a bench run, merged over the full record bench.py wrote beside it (round 6: the line is short, the tables --
    kernel classes, traffic per kernel, in-situ accuracy -- live in `<name>_detail.json`)."""
    line = json.loads(open(path).read().strip().splitlines()[-1])
    det = path.replace(".json", "_detail.json")
    if os.path.exists(det):
        full = json.load(open(det))
        full.update({k: v for k, v in line.items() if k not in full})
        return full
    return line


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:90]


d = last_json(f"{src}/{tag}_bench_n1.json")
json.dump(d, open(f"{dst}/{tag}_bench_n1.json", "w"), indent=1)
for w in ("gru", "composite", "fp32mfma", "bf16x3", "converting"):
    p = f"{src}/{tag}_bench_{w}.json"
    if os.path.exists(p):
        json.dump(last_json(p), open(f"{dst}/{tag}_bench_{w}_informative.json", "w"), indent=1)
for name in (f"{tag}_gemm_pmc.md", f"{tag}_gemm_pmc_fp32mfma.md", f"{tag}_split_accuracy.log", f"{tag}_cpu_baseline_full.json",
             f"{tag}_h2i_accuracy.log", f"{tag}_h2i_timeline.txt"):
    if os.path.exists(f"{src}/{name}"):
        shutil.copy(f"{src}/{name}", f"{dst}/{name}")

# ---- per-shape table
p = f"{src}/{tag}_bench_shapes.json"
if os.path.exists(p) and "kernel_classes" in last_json(p):
    s = last_json(p)
    rows = sorted(((v["ms"], k, v["launches"], v["rate"]) for k, v in s["kernel_classes"].items()), reverse=True)
    tot = sum(r[0] for r in rows)
    gemm = sum(r[0] for r in rows if r[1].startswith("linear_"))
    red = sum(r[0] for r in rows if r[1].startswith("wgrad_reduce"))
    with open(f"{dst}/{tag}_gemm_shapes.md", "w") as f:
        f.write("# Per-shape HIP-event table of one serialised bench step (round %d)\n\n" % rnd)
        f.write("`DTC_PROF_SHAPES=1 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-traffic`; names `linear_*[M x N x K]` "
                "(N = output features, K = input features; `linear_wgrad[M x tiles x layers]` = one grouped weight-gradient launch, "
                "`wgrad_reduce[splits x tiles x layers]` its split reduce).  Rate: TFLOP/s for the GEMM rows, TB/s for the others.  "
                "Times include the ~3 us the two HIP events around a launch add.\n\n")
        f.write("| kernel | launches | ms / step | us / launch | rate |\n|---|---|---|---|---|\n")
        for ms, k, n, rate in rows:
            if ms >= 0.01:
                f.write(f"| `{k}` | {n} | {ms:.3f} | {ms / n * 1e3:.1f} | {rate:.2f} |\n")
        f.write(f"\nTotal {tot:.2f} ms; GEMM launches {gemm:.2f} ms + split reduce {red:.2f} ms; {sum(r[2] for r in rows)} launches per step; "
                f"the same run's overlapped timed region: {s['ms_per_step']:.2f} ms per step.\n")

# ---- traffic table
r = d["roofline"]
if r.get("traffic_kernels"):
    with open(f"{dst}/{tag}_gemm_traffic_step.md", "w") as f:
        f.write("# HBM-side traffic of the GEMM family over one serialised bench step (round %d)\n\n" % rnd)
        f.write(r.get("traffic_source", "") + "\n\nCollected live by `bench.py` (two child runs under `rocprofv3 --kernel-trace --pmc <counter>`, "
                "`tools/analysis/traffic.py`).\n\n| kernel | launches / step | traffic / step (MB) |\n|---|---|---|\n")
        for k, v in sorted(r["traffic_kernels"].items()):
            f.write(f"| `{k}` | {v['launches']} | {v['MB']:.1f} |\n")
        f.write(f"\n* measured: **{r['traffic_step_bytes'] / 1e9:.1f} GB per step** = {r['traffic'] / 1e6:.1f} MB per launch ({r['traffic_launches']} launches)\n")
        f.write(f"* algorithmic (every operand read once, every output written once; in-library accounting): **{r['traffic_algorithmic_step_bytes'] / 1e9:.1f} GB per step** "
                f"= {r['traffic_algorithmic'] / 1e6:.1f} MB per launch\n* ratio **{r['traffic_over_algorithmic']:.3f}**\n\n")
        f.write("Where the excess comes from: the partial slabs of the weight gradients (written once by the grouped kernel, read once by its "
                "reduce kernel -- not counted as algorithmic), the weight images (round 5: 4 bytes per weight + block exponents, written once per phase and fetched once by "
                "every XCD's L2), the one-off packing of the rollout rows into operand images (read 4 B, write 4 B per element, once per mini-batch and update) and operand re-reads that miss the 4 MB per-XCD L2.\n")

# ---- rocprofv3 kernel stats
for mode in ("serial", "overlap"):
    fs = glob.glob(f"{src}/rp_{mode}/**/*kernel_stats.csv", recursive=True)
    if not fs:
        continue
    rows = list(csv.DictReader(open(fs[0])))
    with open(f"{dst}/{tag}_bench_kernel_stats_{mode}.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for x in rows:
            w.writerow([short(x["Name"]), x["Calls"], x["TotalDurationNs"], x["AverageNs"], x["Percentage"], x["MinNs"], x["MaxNs"], x["StdDev"]])
    fam = [x for x in rows if any(k in x["Name"] for k in ("linear_fwd", "linear_dgrad", "linear_wgrad", "wgrad_group", "wgrad_reduce", "gru_step_fwd", "linear_s3", "wgrad_s3", "linear_h2i", "chain_h2i", "wgrad_h2i", "h2i_pack", "h2i_wpack"))]
    tot, calls = sum(float(x["TotalDurationNs"]) for x in fam), sum(int(x["Calls"]) for x in fam)
    print(f"{mode}: GEMM family {tot / 1e6:.1f} ms over {calls} launches -> {tot / calls / 1e3:.2f} us per launch")

for name in (f"{tag}_dp_rehearsal.log", f"{tag}_soak.log", f"{tag}_images_ab.log", f"{tag}_image_kernels.log", f"{tag}_i3_pmc.txt", f"{tag}_wgrad_pmc.txt",
             f"{tag}_h2_power.txt", f"{tag}_energy.txt"):
    if os.path.exists(f"{src}/{name}"):
        shutil.copy(f"{src}/{name}", f"{dst}/{name}")
print("bench:", d["value"], d["ms_per_step"], "roofline", r["achieved"], r["frac"], "traffic ratio", r.get("traffic_over_algorithmic"))
