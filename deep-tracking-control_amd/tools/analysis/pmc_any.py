"""rocprofv3 PMC passes over an arbitrary command; per-kernel averages of every counter (kernels matching a substring).

    python pmc_any.py <out_dir> <kernel substring> -- <command ...>

Passes (SQ: 8 counters each; never combined with tracing domains other than --kernel-trace): the two of gemm_pmc.py."""
import csv
import os
import shutil
import subprocess
import sys

PASSES = {
    "a": "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU",
    "b": "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE",
    "c": "SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT",
}


def main():
    out, sub = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    agg = {}
    for tag, counters in PASSES.items():
        d = os.path.join(os.path.abspath(out), "pmc_" + tag)
        shutil.rmtree(d, ignore_errors=True)
        os.makedirs(d, exist_ok=True)
        full = ["rocprofv3", "--kernel-trace", "--pmc", *counters.split(), "-d", d, "-o", "pmc", "--output-format", "csv", "--", *cmd]
        r = subprocess.run(full, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        if r.returncode != 0:
            print(f"pass {tag} failed: {r.stderr.decode()[-300:]}")
            continue
        path = trace = None
        for base, _, files in os.walk(d):
            for f in files:
                if f.endswith("counter_collection.csv"):
                    path = os.path.join(base, f)
                if f.endswith("kernel_trace.csv"):
                    trace = os.path.join(base, f)
        dur = {}
        if trace:
            with open(trace, newline="") as fh:
                for row in csv.DictReader(fh):
                    dur[row["Dispatch_Id"]] = float(row["End_Timestamp"]) - float(row["Start_Timestamp"])
        per = {}
        with open(path, newline="") as fh:
            for row in csv.DictReader(fh):
                if sub not in row["Kernel_Name"]:
                    continue
                name = row["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "")
                key = (name.split("(")[0][:60], row.get("Grid_Size"))
                e = per.setdefault((key, row["Dispatch_Id"]), {})
                e[row["Counter_Name"]] = e.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
                if tag == "a" and row["Dispatch_Id"] in dur:
                    e["_duration_ns"] = dur[row["Dispatch_Id"]]
        for (key, _), c in per.items():
            a = agg.setdefault(key, {"_n": {}})
            for k, v in c.items():
                a[k] = a.get(k, 0.0) + v
                a["_n"][k] = a["_n"].get(k, 0) + 1
    for key, a in agg.items():
        n = a.pop("_n")
        print(f"== {key[0]} grid {key[1]}  ({max(n.values())} dispatches)")
        avg = {k: v / n[k] for k, v in a.items()}
        for k in sorted(avg):
            print(f"   {k:34s} {avg[k]:16.0f}")
        g = avg.get("GRBM_GUI_ACTIVE")
        if g:
            simd_cycles = 1024.0 * g / 8.0
            if avg.get("_duration_ns"):
                print(f"   -> kernel duration {avg['_duration_ns'] / 1e3:.1f} us (counter pass a), clock = GRBM_GUI_ACTIVE / 8 XCDs / duration = {g / 8.0 / avg['_duration_ns']:.2f} GHz")
            print(f"   -> MFMA busy {avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / simd_cycles:.3f};  per wave-cycle: parked {avg.get('SQ_WAIT_ANY', 0) / max(1, avg.get('SQ_WAVE_CYCLES', 1)):.3f}, "
                  f"issue-stalled {avg.get('SQ_WAIT_INST_ANY', 0) / max(1, avg.get('SQ_WAVE_CYCLES', 1)):.3f}, issuing {avg.get('SQ_ACTIVE_INST_ANY', 0) / max(1, avg.get('SQ_WAVE_CYCLES', 1)):.3f}")
        if avg.get("SQ_LDS_IDX_ACTIVE"):
            print(f"   -> LDS: conflict cycles / active cycles {avg.get('SQ_LDS_BANK_CONFLICT', 0) / avg['SQ_LDS_IDX_ACTIVE']:.3f}")


if __name__ == "__main__":
    main()
