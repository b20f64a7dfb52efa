# what bounds linear_h2i_kernel's K loop: 24576 x 512 x 512 forward / data gradient, image -> image, back to back, with pieces of the stage removed
#   2: no MFMA   4: no fragment reads   8: no LDS-DMA transfers (results are wrong, times are the point)
O=gpurun_out; mkdir -p $O; : > $O/r06_h2i_ablate.txt
B=deep-tracking-control_amd/tools/_bin
for rnd in 1 2; do for t in 0 2 4 8 12 14; do
  lib=$B/libdtc_hip_habl$t.so; [ $t = 0 ] && lib=deep-tracking-control_amd/dtc_amd/lib/libdtc_hip.so
  echo "== ablation $t: $(DTC_LIB=$PWD/$lib timeout 200 python deep-tracking-control_amd/tools/h2i_probe.py all time 2>&1 | grep -v amdgpu.ids | tr '\n' ' ')" >> $O/r06_h2i_ablate.txt
done; done
cat $O/r06_h2i_ablate.txt
