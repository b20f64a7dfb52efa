"""Analyse a rocprofv3 --kernel-trace CSV of bench.py: per timed step (foothold_plan_kernel marks the start) the
wall span, union-busy time, idle gaps, concurrency histogram and per-queue busy time.
    python timeline.py <kernel_trace.csv> [step_index]"""
import csv
import sys
import collections

rows = list(csv.DictReader(open(sys.argv[1])))
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"]))) for r in rows]
ev.sort()
starts = [i for i, e in enumerate(ev) if "foothold_plan" in e[2]]
print("steps found:", len(starts))
which = int(sys.argv[2]) if len(sys.argv) > 2 else 1
a, b = starts[which], starts[which + 1]
step = ev[a:b]
t0 = step[0][0]
t1 = max(e[1] for e in step)
print(f"step {which}: {len(step)} dispatches, span {(t1 - t0) / 1e6:.2f} ms")
# sweep
pts = []
for s, e, n, q, g in step:
    pts.append((s, 1, n))
    pts.append((e, -1, n))
pts.sort()
cur = 0
last = t0
hist = collections.Counter()
for t, d, n in pts:
    hist[cur] += t - last
    last = t
    cur += d
tot = sum(hist.values())
for k in sorted(hist):
    print(f"  concurrency {k}: {hist[k] / 1e6:8.3f} ms ({100 * hist[k] / tot:5.1f} %)")
# per queue busy
perq = collections.defaultdict(float)
for s, e, n, q, g in step:
    perq[q] += e - s
for q, v in sorted(perq.items()):
    print(f"  queue {q}: kernel time {v / 1e6:8.3f} ms")
# time with at least one 'wide' GEMM (grid >= 700 blocks) running
wide = [(s, e) for s, e, n, q, g in step if g >= 700 and ("linear_" in n or "wgrad_group_kernel" in n)]
wide.sort()
u, ce = 0, None
for s, e in wide:
    if ce is None or s > ce:
        u += e - s
        ce = e
    elif e > ce:
        u += e - ce
        ce = e
print(f"  union time of >=700-block GEMM kernels: {u / 1e6:.3f} ms; their summed durations {sum(e - s for s, e in wide) / 1e6:.3f} ms")
by = collections.defaultdict(lambda: [0, 0.0])
for s, e, n, q, g in step:
    k = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40]
    by[k][0] += 1
    by[k][1] += e - s
for k, (c, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:25]:
    print(f"  {k:42s} {c:5d} {t / 1e6:8.3f} ms")

# ---- exposed time: intervals in which no >=700-block GEMM runs; which kernels run there
wide.sort()
merged = []
for s, e in wide:
    if merged and s <= merged[-1][1]:
        merged[-1][1] = max(merged[-1][1], e)
    else:
        merged.append([s, e])
gaps = []
prev = t0
for s, e in merged:
    if s > prev:
        gaps.append((prev, s))
    prev = max(prev, e)
if prev < t1:
    gaps.append((prev, t1))
tot_gap = sum(e - s for s, e in gaps)
print(f"  time with no wide GEMM in flight: {tot_gap / 1e6:.3f} ms in {len(gaps)} intervals")
acc = collections.defaultdict(float)
for s, e, n, q, g in step:
    if g >= 700 and ("linear_" in n or "wgrad_group_kernel" in n):
        continue
    for gs, ge in gaps:
        lo, hi = max(s, gs), min(e, ge)
        if hi > lo:
            acc[n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:40] + f" g{g}"] += hi - lo
for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:30]:
    print(f"    {k:50s} {v / 1e6:7.3f} ms")

# time inside those intervals in which nothing runs at all (dispatch gaps, cross-stream event waits)
busy = []
for s, e, n, q, g in step:
    if g >= 700 and ("linear_" in n or "wgrad_group_kernel" in n):
        continue
    for gs, ge in gaps:
        lo, hi = max(s, gs), min(e, ge)
        if hi > lo:
            busy.append((lo, hi))
busy.sort()
u, ce = 0, None
for s, e in busy:
    if ce is None or s > ce:
        u += e - s
        ce = e
    elif e > ce:
        u += e - ce
        ce = e
print(f"  of which some other kernel runs: {u / 1e6:.3f} ms; nothing runs: {(tot_gap - u) / 1e6:.3f} ms")
