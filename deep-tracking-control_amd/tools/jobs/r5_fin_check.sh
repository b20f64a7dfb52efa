timeout 1800 python -m pytest tests/test_hip_ppo.py tests/test_hip_kernels.py tests/test_gru_path.py tests/test_composite_path.py -x -q -m gpu 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/rp_fin
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/rp_fin -o bench --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-traffic --no-in-situ > /dev/null 2>&1
grep -h "ppo_loss_finalize\|ppo_heads_loss\|h2i_wpack" $(find /tmp/rp_fin -name "*kernel_stats.csv") | cut -c1-120
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', round(d['ms_per_step'],3), round(d['value']))"; done
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc
