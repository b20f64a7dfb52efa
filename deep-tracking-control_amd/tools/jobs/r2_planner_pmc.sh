# PMC counters of the planner kernel (98 304 maps): instruction mix and VALU busy -> gpurun_out/r2/planner_pmc.txt
mkdir -p gpurun_out/r2
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for C in "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_WAIT_ANY" "FETCH_SIZE" "WRITE_SIZE"; do
i=$((i+1))
rm -rf /tmp/pmc_pl_$i
timeout 300 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_pl_$i -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py scorer > /dev/null 2>&1
done
python - <<'PY' | tee $R/gpurun_out/r2/planner_pmc.txt
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/pmc_pl_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'foothold_plan' in r['Kernel_Name']:
            acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = v[2:] if len(v) > 4 else v
        print(f"  {c:24s} {sum(v)/len(v):16.1f}   per env {sum(v)/len(v)/98304:10.2f}")
PY
