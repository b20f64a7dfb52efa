# Round 3: first contact of the split-precision GEMM path: accuracy tests, timing, the PPO parity suite with the path on
O=gpurun_out/r3
mkdir -p $O
timeout 900 python -m pytest tests/test_hip_split.py -m gpu -x -q -s 2>&1 | tail -40 > $O/split1.log
cat $O/split1.log
DTC_GEMM_SPLIT=1 timeout 1200 python -m pytest tests/test_hip_ppo.py -m gpu -q -k "teacher_forced_64 or strict or fused_heads or forward_act" 2>&1 | tail -15 > $O/split1_ppo.log
cat $O/split1_ppo.log
for v in 0 1; do
echo -n "DTC_GEMM_SPLIT=$v: "
DTC_GEMM_SPLIT=$v timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']), round(d['roofline']['frac'],4), d['last_update'][:3])"
done | tee $O/ab_split1.log
