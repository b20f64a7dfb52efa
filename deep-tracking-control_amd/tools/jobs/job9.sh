mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $R/gpurun_out/pmc_sc1 -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py scorer > $R/gpurun_out/pmc_sc1.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32 GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_sc2 -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py scorer > $R/gpurun_out/pmc_sc2.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/pmc_sc3 -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py scorer > $R/gpurun_out/pmc_sc3.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/pmc_sc4 -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py scorer > $R/gpurun_out/pmc_sc4.log 2>&1
ls $R/gpurun_out/pmc_sc1 $R/gpurun_out/pmc_sc3
