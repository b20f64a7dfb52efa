#!/bin/bash
# rocprofv3 kernel stats of a short bench run on the two-term fp16 path (top kernels by total time)
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r4h2
mkdir -p $OUT
rm -rf /tmp/rp_h2
rocprofv3 --kernel-trace --stats -d /tmp/rp_h2 -o h2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-traffic > $OUT/bench_under_rocprof.json 2> $OUT/rocprof.err
f=$(find /tmp/rp_h2 -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY' > $OUT/h2_kernel_stats.txt
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time {tot/1e6:.1f} ms over 6 steps")
for r in rows[:28]:
    n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "")
    print(f"{n[:100]:100s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:8.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms  {float(r['Percentage']):5.1f} %")
PY
cat $OUT/h2_kernel_stats.txt
