# variants of wgrad_h2i_group_kernel alone on the chip (tools/build_variant.sh <tag> wgrad_h2i.hip -D...): us per grouped launch (+ reduce)
O=gpurun_out; mkdir -p $O; : > $O/r06_wgrad_var.txt
B=deep-tracking-control_amd/tools/_bin
for rnd in 1 2; do for t in base $@; do
  lib=$B/libdtc_hip_$t.so; [ $t = base ] && lib=deep-tracking-control_amd/dtc_amd/lib/libdtc_hip.so
  echo "== $t" >> $O/r06_wgrad_var.txt
  DTC_LIB=$PWD/$lib timeout 200 python deep-tracking-control_amd/tools/wgrad_probe.py 2>&1 | grep tiles >> $O/r06_wgrad_var.txt
done; done
cat $O/r06_wgrad_var.txt
