# weight-gradient batch-slice count: fewer, longer blocks write fewer partial slabs (HBM traffic) -- where does the time stop being neutral?
run() { env "$@" timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-traffic 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', round(d['ms_per_step'],3), round(d['value']))"; }
run A=0
for b in 512 768 1024 1280 1536; do run DTC_WGRAD_S3_BLOCKS=$b; done
run A=0
