mkdir -p gpurun_out/r4e
cd /tmp && export TMPDIR=/tmp
python $GRAFT_REPO_ROOT/deep-tracking-control_amd/tools/analysis/pmc_any.py $GRAFT_REPO_ROOT/gpurun_out/r4e/i3 linear_i3_kernel -- python $GRAFT_REPO_ROOT/deep-tracking-control_amd/tools/i3_ablate.py product > $GRAFT_REPO_ROOT/gpurun_out/r4e/pmc_i3.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r4e/pmc_i3.txt | grep -v amdgpu.ids | tail -60
