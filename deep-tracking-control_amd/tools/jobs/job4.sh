mkdir -p gpurun_out
R=$PWD
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -15 > gpurun_out/t4.log; cat gpurun_out/t4.log
timeout 200 python deep-tracking-control_amd/tools/microbench.py gemm > gpurun_out/mb3.log 2>&1; cat gpurun_out/mb3.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\b(SQ_[A-Z0-9_]+|TCC_[A-Z0-9_]+|GRBM_[A-Z_]+|FETCH_SIZE|WRITE_SIZE)\b" | sort -u > $R/gpurun_out/counters.txt; wc -l $R/gpurun_out/counters.txt
for T in fwd wgrad; do
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE -d $R/gpurun_out/pmc_$T -o pmc --output-format csv -- python $R/deep-tracking-control_amd/tools/prof_target.py $T > $R/gpurun_out/pmc_$T.log 2>&1
tail -2 $R/gpurun_out/pmc_$T.log
done
ls -R $R/gpurun_out/pmc_fwd | head -20
