# narrow (fp32 single-pass) grouped weight gradients: batch slices per launch (DTC_WGRAD_GROUP_BLOCKS / tiles); two interleaved rounds
for r in 1 2; do
for b in 3072 1536 768 384 192; do
echo -n "DTC_WGRAD_GROUP_BLOCKS=$b: "
DTC_WGRAD_GROUP_BLOCKS=$b python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2))"
done
done
