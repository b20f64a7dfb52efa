# K-loop ablation of linear_i3_kernel -> gpurun_out/r4p/r04_i3_ablation.txt (variant libraries: tools/build_variant.sh i3p<mask> gemm_s3.hip ... -DDTC_I3_PROBE=<mask>)
T=deep-tracking-control_amd/tools
O=gpurun_out/r4p
mkdir -p $O
( echo "linear_i3_kernel, forward 24576 x 512 x 512 + ReLU, fp32 result; DTC_I3_PROBE bits: 1 no LDS-DMA in the loop, 2 no fragment reads, 4 no barrier,"
  echo "8 no stores, 16 every LDS-DMA lands in buffer 1 while every fragment is read from buffer 0 (traffic without data dependence)."
  echo "NOTE: every probe that cuts the operand path computes on STATIC operand bits -- it also measures the zero-operand clock (profiles/r04_clock.txt)."
  python $T/i3_ablate.py "product" 2>/dev/null
  python $T/i3_ablate.py "product" image 2>/dev/null
  for m in 1 2 3 4 7 8 15 16 20; do DTC_LIB=$T/_bin/libdtc_hip_i3p$m.so python $T/i3_ablate.py "probe $m" 2>&1 | grep -v amdgpu | tail -1; done
  DTC_LIB=$T/_bin/libdtc_hip_i3mid.so python $T/i3_ablate.py "LDS-DMA issued under the first six MFMAs of a stage" 2>&1 | grep -v amdgpu | tail -1
  python $T/i3_ablate.py "product (again)" 2>/dev/null ) > $O/r04_i3_ablation.txt 2>&1
cat $O/r04_i3_ablation.txt
