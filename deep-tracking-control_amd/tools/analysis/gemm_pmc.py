"""SQ / GRBM counters of the GEMM-family kernels over one serialised bench step (rocprofv3 PMC, --kernel-trace only).

    python gemm_pmc.py collect <out_dir> [workload]    two counter passes over `bench.py --steps 1 --warmup 0` + summary (markdown)
    python gemm_pmc.py parse <out_dir>                 re-read existing passes

Per kernel (template instance): launches, kernel time (rocprofv3 start -> end of the same pass), MFMA-pipe busy fraction
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs)
the EXECUTED matrix rate (every v_mfma_f32_32x32x2_f32 = 64 busy cycles = 4096 FLOP, padding included; the split-precision
kernels: fp32-equivalent FLOP, i.e. bf16 FLOP / 6) and the
instruction mix per MFMA.  `bench.py` imports `collect()` for `roofline.mfma_busy`.
"""
import csv
import json
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(HERE)))
PASSES = {
    "a": "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
    "b": "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR",
}
BY_GRID = os.environ.get("DTC_PMC_BY_GRID", "1") != "0"      # one row per (kernel, workgroup count) = per layer shape
FAMILY = ("linear_fwd_kernel", "linear_dgrad_kernel", "linear_wgrad_kernel", "wgrad_group_kernel", "wgrad_reduce_kernel",
          "wgrad_group_reduce_kernel", "gru_step_fwd_kernel", "linear_s3_kernel", "wgrad_s3_group_kernel", "wgrad_s3_reduce_kernel", "wimage_kernel",
          "linear_h2i_kernel", "chain_h2i_kernel", "wgrad_h2i_group_kernel", "wgrad_h2i_reduce_kernel", "h2i_pack_kernel", "h2i_wpack_kernel",
          "gru_s3_kernel")


def short(name):
    if not any(f in name for f in FAMILY):
        return None
    n = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    return n[:60]


def run_pass(tag, out_dir, workload="decoder", timeout=900):
    d = os.path.join(os.path.abspath(out_dir), "pmc_" + tag)       # rocprofv3 runs with cwd = /tmp
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp", DTC_OVERLAP_WGRAD="0", DTC_OVERLAP_LANES="0")
    cmd = ["rocprofv3", "--kernel-trace", "--pmc", *PASSES[tag].split(), "-d", d, "-o", "pmc", "--output-format", "csv", "--",
           sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-traffic",
           "--workload", workload]
    r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"rocprofv3 pass {tag} failed ({r.returncode}): {r.stderr.decode()[-400:]}")
    return d


def _find(d, suffix):
    for base, _, files in os.walk(d):
        for f in files:
            if f.endswith(suffix):
                return os.path.join(base, f)
    raise RuntimeError(f"no *{suffix} under {d}")


def parse_pass(d):
    """{kernel: {counter: sum over the last step, "_n": launches, "_ns": kernel time}} for the last (serialised) step."""
    rows = {}
    with open(_find(d, "counter_collection.csv"), newline="") as fh:
        for r in csv.DictReader(fh):
            e = rows.setdefault(int(r["Dispatch_Id"]), dict(name=r["Kernel_Name"], c={}, grid=int(r.get("Grid_Size", 0) or 0) // 256))
            e["c"][r["Counter_Name"]] = e["c"].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
            if "Start_Timestamp" in r and r.get("End_Timestamp"):
                e["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    try:
        with open(_find(d, "kernel_trace.csv"), newline="") as fh:
            for r in csv.DictReader(fh):
                e = rows.get(int(r["Dispatch_Id"]))
                if e is not None:
                    e["ns"] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    except RuntimeError:
        pass
    ids = sorted(rows)
    starts = [i for i in ids if "foothold_plan" in rows[i]["name"]]
    if not starts:
        raise RuntimeError("no foothold_plan dispatch in the trace")
    out = {}
    for i in ids:
        if i < starts[-1]:
            continue
        k = short(rows[i]["name"])
        if not k:
            continue
        if BY_GRID:
            k = f"{k} [{rows[i]['grid']} wg]"
        o = out.setdefault(k, {"_n": 0, "_ns": 0.0})
        o["_n"] += 1
        o["_ns"] += rows[i].get("ns", 0.0)
        for c, v in rows[i]["c"].items():
            o[c] = o.get(c, 0.0) + v
    return out


def summarise(out_dir):
    out_dir = os.path.abspath(out_dir)
    a, b = parse_pass(os.path.join(out_dir, "pmc_a")), parse_pass(os.path.join(out_dir, "pmc_b"))
    res = {}
    for k in sorted(a):
        ca, cb = a[k], b.get(k, {})
        busy, gui = ca.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), ca.get("GRBM_GUI_ACTIVE", 0.0)
        # single-pass kernels: v_mfma_f32_32x32x2_f32 = 64 busy cycles, 4096 FLOP; split kernels (_s3_): v_mfma_f32_32x32x16_bf16 =
        # 32 busy cycles, 32768 bf16 FLOP = 32768 / 6 fp32-equivalent FLOP (six passes per product)
        # operand-image kernels (_h2i_): v_mfma_f32_32x32x16_f16, three passes per product
        h2 = "_h2i_" in k
        s3 = "_s3_" in k or h2
        n_mfma = busy / (32.0 if s3 else 64.0)
        sec = ca["_ns"] * 1e-9
        res[k] = dict(launches=ca["_n"], ms=ca["_ns"] * 1e-6, mfma_busy=busy / (1024.0 * gui / 8.0) if gui else 0.0,
                      executed_tflops=n_mfma * (32768.0 / 3.0 if h2 else 32768.0 / 6.0 if s3 else 4096.0) / sec / 1e12 if sec > 0 else 0.0,
                      clock_ghz=gui / 8.0 / (ca["_ns"]) if ca["_ns"] else 0.0,
                      valu_per_mfma=cb.get("SQ_INSTS_VALU", 0.0) / n_mfma - 1.0 if n_mfma else 0.0,
                      salu_per_mfma=cb.get("SQ_INSTS_SALU", 0.0) / n_mfma if n_mfma else 0.0,
                      lds_per_mfma=cb.get("SQ_INSTS_LDS", 0.0) / n_mfma if n_mfma else 0.0,
                      vmem_rd_per_mfma=cb.get("SQ_INSTS_VMEM_RD", 0.0) / n_mfma if n_mfma else 0.0,
                      lds_conflict_frac=cb.get("SQ_LDS_BANK_CONFLICT", 0.0) / cb["SQ_ACTIVE_INST_LDS"] if cb.get("SQ_ACTIVE_INST_LDS") else 0.0,
                      wait_inst_lds_frac=cb.get("SQ_WAIT_INST_LDS", 0.0) / ca["SQ_WAVE_CYCLES"] if ca.get("SQ_WAVE_CYCLES") else 0.0,
                      wait_any_frac=ca.get("SQ_WAIT_ANY", 0.0) / ca["SQ_WAVE_CYCLES"] if ca.get("SQ_WAVE_CYCLES") else 0.0,
                      wait_inst_any_frac=ca.get("SQ_WAIT_INST_ANY", 0.0) / ca["SQ_WAVE_CYCLES"] if ca.get("SQ_WAVE_CYCLES") else 0.0)
    tot_busy = sum(a[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for k in a)
    tot_gui = sum(a[k].get("GRBM_GUI_ACTIVE", 0.0) for k in a)
    return dict(kernels=res, family_mfma_busy=tot_busy / (1024.0 * tot_gui / 8.0) if tot_gui else 0.0,
                family_ms=sum(a[k]["_ns"] for k in a) * 1e-6, family_launches=sum(a[k]["_n"] for k in a))


def markdown(s):
    lines = ["| kernel | launches | ms / step | MFMA busy | executed TFLOP/s | clock GHz | VALU / MFMA | SALU / MFMA | LDS instr / MFMA | VMEM rd / MFMA | LDS conflict / LDS busy | wait-LDS-issue | wait-any | wait-issue-any |",
             "|---|---|---|---|---|---|---|---|---|---|---|---|---|---|"]
    for k, v in sorted(s["kernels"].items(), key=lambda kv: -kv[1]["ms"]):
        lines.append(f"| `{k}` | {v['launches']} | {v['ms']:.2f} | {v['mfma_busy']:.3f} | {v['executed_tflops']:.1f} | {v['clock_ghz']:.2f} | "
                     f"{v['valu_per_mfma']:.2f} | {v['salu_per_mfma']:.2f} | {v['lds_per_mfma']:.2f} | {v['vmem_rd_per_mfma']:.3f} | "
                     f"{v['lds_conflict_frac']:.2f} | {v['wait_inst_lds_frac']:.3f} | {v['wait_any_frac']:.2f} | {v['wait_inst_any_frac']:.2f} |")
    lines.append("")
    lines.append(f"Family: {s['family_launches']} launches, {s['family_ms']:.1f} ms under the profiler, MFMA busy {s['family_mfma_busy']:.3f}.")
    return "\n".join(lines)


def collect(out_dir, workload="decoder"):
    if shutil.which("rocprofv3") is None:
        raise RuntimeError("rocprofv3 not on PATH")
    for t in PASSES:
        run_pass(t, out_dir, workload)
    summary = summarise(out_dir)
    with open(os.path.join(out_dir, "summary.json"), "w") as fh:      # the raw rocprofv3 trees are tens of MB per pass: keep the summary only
        json.dump(summary, fh, indent=1)
    if os.environ.get("DTC_KEEP_PMC") != "1":
        for d in os.listdir(out_dir):
            if os.path.isdir(os.path.join(out_dir, d)):
                shutil.rmtree(os.path.join(out_dir, d), ignore_errors=True)
    return summary


if __name__ == "__main__":
    mode, out = sys.argv[1], sys.argv[2]
    s = collect(out, *(sys.argv[3:4])) if mode == "collect" else summarise(out)
    print(markdown(s))
    print()
    print(json.dumps({k: v for k, v in s.items() if k != "kernels"}))
