mkdir -p gpurun_out
T=deep-tracking-control_amd/tools
python $T/h2i_probe.py all time > gpurun_out/r5_h2i_time.txt 2>&1
python $T/h2i_probe.py all time zero >> gpurun_out/r5_h2i_time.txt 2>&1
python $T/analysis/pmc_any.py gpurun_out/r5_pmc_h2i h2i -- python $PWD/$T/h2i_probe.py all > gpurun_out/r5_h2i_pmc.txt 2>&1
cat gpurun_out/r5_h2i_time.txt gpurun_out/r5_h2i_pmc.txt
