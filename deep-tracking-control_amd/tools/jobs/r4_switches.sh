# the bench step under the runtime switches with the fp16 kernels as the default: nothing may crash, the losses stay finite; A/B of the
# two switches that looked neutral-or-better in the first pass
run() { env "$@" python bench.py --steps ${STEPS:-3} --warmup 2 --no-cpu-baseline --no-traffic 2>/tmp/err.txt | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-40s %.0f env-steps/s  %.2f ms  loss %s' % ('$*', d['value'], d['ms_per_step'], [round(x, 4) for x in d.get('last_update')][:3]))" || tail -3 /tmp/err.txt; }
run DTC_S3_WIMG=0
run DTC_WIMG_GROUP=0
for rep in 1 2; do
  STEPS=10 run DTC_NOP=1
  STEPS=10 run DTC_PACK_INPUTS=0
  STEPS=10 run DTC_FUSE_HEADS=0
  STEPS=10 run DTC_PACK_INPUTS=0 DTC_FUSE_HEADS=0
done
