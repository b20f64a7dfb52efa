# planner register / occupancy sweep: variant libraries (tools/_bin/libdtc_hip_fh<k>.so, built by tools/build_planner_variants.sh)
mkdir -p gpurun_out/r2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for k in ${FH_VARIANTS:-5 6 7}; do
  L=$R/deep-tracking-control_amd/tools/_bin/libdtc_hip_fh$k.so
  [ -f $L ] || continue
  rm -rf /tmp/rp_$k
  DTC_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_$k -o p --output-format csv -- python $R/deep-tracking-control_amd/tools/planner_time.py 2>/dev/null | grep "us per call"
  f=$(find /tmp/rp_$k -name "*kernel_stats.csv" | head -1)
  python - "$f" "$k" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "foothold" in r["Name"]:
        print(f"variant {sys.argv[2]}: {r['Name'][:60]:60s} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f} max {float(r['MaxNs'])/1e3:.1f}")
PY
done 2>&1 | tee $R/gpurun_out/r2/planner_sweep.txt
