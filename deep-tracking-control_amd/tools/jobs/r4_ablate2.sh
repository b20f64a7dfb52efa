T=deep-tracking-control_amd/tools
for i in 1 2; do
python $T/i3_ablate.py "product" 2>/dev/null
DTC_LIB=$T/_bin/libdtc_hip_i3mid.so python $T/i3_ablate.py "dma mid-stage" 2>/dev/null
done
