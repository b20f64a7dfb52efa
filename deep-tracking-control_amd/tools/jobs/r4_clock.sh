# the clock the image-operand forward kernel runs at on random vs zero operands (GRBM_GUI_ACTIVE / duration), and the sustained MFMA
# rate for three orders of the six plane pairs of a product
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r4p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
( for m in random zero; do echo "== linear_i3_kernel, 24576 x 512 x 512, $m operands"; python $R/deep-tracking-control_amd/tools/analysis/pmc_any.py $O/pmc_clk linear_i3_kernel -- python $R/deep-tracking-control_amd/tools/i3_power.py $m 2>&1 | grep -E "==|GRBM_GUI|MFMA_BUSY|WAVE_CYCLES|->"; done ) > $O/r04_clock.txt 2>&1
cd $R
( for o in 0 1 2; do DTC_PROBE_ORDER=$o python - <<'PY'
import os, sys
sys.path.insert(0, 'deep-tracking-control_amd')
from dtc_amd import ops
print('pass order', os.environ['DTC_PROBE_ORDER'], ': sustained TFLOP/s fp32-equivalent  zero operands %.1f   random operands %.1f' % (ops.mfma_sustained('cuda:0', False), ops.mfma_sustained('cuda:0', True)))
PY
done ) >> $O/r04_clock.txt 2>&1
rm -rf $O/pmc_clk
cat $O/r04_clock.txt | grep -v amdgpu.ids
