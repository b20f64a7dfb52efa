mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
rm -rf $R/gpurun_out/pmc_tr_$C
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_tr_$C -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py traffic > /dev/null 2>&1
done
ls $R/gpurun_out/pmc_tr_FETCH_SIZE
