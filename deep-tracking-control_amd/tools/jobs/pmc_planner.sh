mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
i=0
for C in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_WAVES GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"; do
i=$((i+1))
rm -rf $R/gpurun_out/pmc_sc_$i
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $R/gpurun_out/pmc_sc_$i -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py scorer > /dev/null 2>&1
done
ls $R/gpurun_out/pmc_sc_1
