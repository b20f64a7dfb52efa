# K-loop ablation of linear_i3_kernel (DTC_I3_PROBE bits: 1 no LDS-DMA, 2 no fragment reads, 4 no barrier, 8 no stores)
T=deep-tracking-control_amd/tools
python $T/i3_ablate.py "product" 2>/dev/null
python $T/i3_ablate.py "product" image 2>/dev/null
for m in 1 2 3 4 7 8 15; do
  DTC_LIB=$T/_bin/libdtc_hip_i3p$m.so python $T/i3_ablate.py "probe $m" 2>/dev/null
done
