# ablation ladder of the split forward kernel's K loop (variants built by the caller into tools/_bin/libdtc_hip_p<mask>.so)
B=$PWD/deep-tracking-control_amd/tools/_bin
for m in $MASKS; do
DTC_LIB=$B/libdtc_hip_p$m.so DTC_SKIP_ABI_CHECK=1 timeout 120 python deep-tracking-control_amd/tools/s3_ablate.py "probe mask $m" 2>&1 | tail -1
done
