# planner decomposition sweep: envs per wave (0 = persistent split) -> gpurun_out/r2/planner_epw.txt
mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for e in ${FH_EPW:-0 4 8 12 16 24}; do
  rm -rf /tmp/rp_e$e
  DTC_FH_EPW=$e timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_e$e -o p --output-format csv -- python $R/deep-tracking-control_amd/tools/planner_time.py > /dev/null 2>&1
  f=$(find /tmp/rp_e$e -name "*kernel_stats.csv" | head -1)
  python - "$f" "$e" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "foothold" in r["Name"]:
        print(f"envs/wave {sys.argv[2]:>3s}: calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f} max {float(r['MaxNs'])/1e3:.1f}")
PY
done 2>&1 | tee $R/gpurun_out/r2/planner_epw.txt
