# what bounds wgrad_h2i_group_kernel: the kernel alone on the chip with pieces of its stage removed (results are wrong, times are the point)
#   1: no row factors (v_pk_mul_f16)   2: no MFMA   4: no fragment reads (ds_read_b64_tr_b16)   8: no LDS-DMA transfers
O=gpurun_out; mkdir -p $O; : > $O/r06_wgrad_ablate.txt
B=deep-tracking-control_amd/tools/_bin
for t in 0 1 2 4 8 12 14; do
  lib=$B/libdtc_hip_abl$t.so; [ $t = 0 ] && lib=deep-tracking-control_amd/dtc_amd/lib/libdtc_hip.so
  echo "== ablation $t" >> $O/r06_wgrad_ablate.txt
  DTC_LIB=$PWD/$lib timeout 200 python deep-tracking-control_amd/tools/wgrad_probe.py 2>&1 | grep tiles | sed 's/four buffers.*rounds/rounds/' >> $O/r06_wgrad_ablate.txt
done
cat $O/r06_wgrad_ablate.txt
