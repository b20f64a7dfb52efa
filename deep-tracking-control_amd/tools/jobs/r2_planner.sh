# planner iteration: exhaustive sqrt / division probes, scorer parity tests, planner kernel time -> gpurun_out/r2/
mkdir -p gpurun_out/r2 /tmp/pb
R=$PWD
cd deep-tracking-control_amd
F="--offload-arch=gfx950 -O3 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt"
if [ "${PROBES:-1}" = "1" ]; then
/opt/rocm/bin/hipcc $F -I csrc tools/probes/sqrt_exact.hip -o /tmp/pb/sqrt_exact 2>/dev/null && timeout 300 /tmp/pb/sqrt_exact | tee $R/gpurun_out/r2/sqrt_exact.txt
/opt/rocm/bin/hipcc $F tools/probes/div_const_exact.hip -o /tmp/pb/div_const_exact 2>/dev/null && timeout 300 /tmp/pb/div_const_exact | tee $R/gpurun_out/r2/div_const_exact.txt
fi
cd $R
timeout 1200 python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "scorer or heights or foothold or plan or patch_env" 2>&1 | tail -15 | tee gpurun_out/r2/planner_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_pl
timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/rp_pl -o p --output-format csv -- python $R/deep-tracking-control_amd/tools/planner_time.py 2>/dev/null | grep "us per call"
python - $(find /tmp/rp_pl -name "*kernel_stats.csv" | head -1) <<'PY' | tee $R/gpurun_out/r2/planner_time.txt
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "foothold" in r["Name"]:
        print(f"{r['Name'][:64]:64s} calls {r['Calls']} avg {float(r['AverageNs'])/1e3:.1f} us min {float(r['MinNs'])/1e3:.1f} max {float(r['MaxNs'])/1e3:.1f}")
PY
