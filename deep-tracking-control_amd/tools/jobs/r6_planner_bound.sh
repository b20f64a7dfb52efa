# upper bound of what compacting the in-radius candidates could buy the planner: legs >= k scored by the distance term alone (wrong results, times only)
O=gpurun_out; mkdir -p $O; : > $O/r06_planner_bound.txt
B=deep-tracking-control_amd/tools/_bin
for rnd in 1 2; do for t in base abl3 abl2 abl1; do
  lib=$B/libdtc_hip_fh$t.so; [ $t = base ] && lib=deep-tracking-control_amd/dtc_amd/lib/libdtc_hip.so
  DTC_LIB=$PWD/$lib timeout 200 python deep-tracking-control_amd/tools/planner_time.py 2>&1 | grep "fast" | sed "s/^/$t: /" >> $O/r06_planner_bound.txt
done; done
cat $O/r06_planner_bound.txt
