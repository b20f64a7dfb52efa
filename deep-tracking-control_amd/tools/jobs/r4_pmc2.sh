mkdir -p gpurun_out/r4g
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/deep-tracking-control_amd/tools/analysis/pmc_any.py $R/gpurun_out/r4g/w wgrad_ -- python $R/deep-tracking-control_amd/tools/img_probe.py wgrad > $R/gpurun_out/r4g/pmc_wgrad.txt 2>&1
grep -v amdgpu.ids $R/gpurun_out/r4g/pmc_wgrad.txt | grep -v "^   SQ_INSTS_SMEM\|SQ_WAVES\|ACTIVE_INST_MISC\|ACTIVE_INST_SCA" | tail -150
