for bn in 128 64; do for bl in 512 640 768 1024 1536 2048; do
DTC_WGRAD_BN=$bn DTC_WGRAD_BLOCKS=$bl timeout 120 python deep-tracking-control_amd/tools/microbench.py wgrad 2>&1 | grep wgrad
done; done
