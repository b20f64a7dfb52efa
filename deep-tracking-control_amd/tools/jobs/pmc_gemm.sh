mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
for T in fwd dgrad; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc2_${T}_a -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py $T > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VMEM -d $R/gpurun_out/pmc2_${T}_b -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py $T > /dev/null 2>&1
done
