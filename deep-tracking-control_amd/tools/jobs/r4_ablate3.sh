T=deep-tracking-control_amd/tools
python $T/i3_ablate.py "product" 2>/dev/null
DTC_LIB=$T/_bin/libdtc_hip_i3p16.so python $T/i3_ablate.py "DMA -> buffer 1, reads <- buffer 0 (no data dependence)" 2>/dev/null
DTC_LIB=$T/_bin/libdtc_hip_i3p20.so python $T/i3_ablate.py "same, no barrier" 2>/dev/null
python $T/i3_ablate.py "product" 2>/dev/null
