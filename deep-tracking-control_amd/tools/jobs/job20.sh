for thr in 256 320; do
echo "MIN_BLOCKS=$thr"
DTC_GEMM_MIN_BLOCKS=$thr timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('decoder', d['value'], d['ms_per_step'])"
DTC_GEMM_MIN_BLOCKS=$thr timeout 300 python bench.py --workload composite --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('composite', d['value'], d['ms_per_step'])"
done
DTC_GEMM_MIN_BLOCKS=0 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('decoder thr0', d['value'], d['ms_per_step'])"
DTC_GEMM_MIN_BLOCKS=512 timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('decoder thr512', d['value'], d['ms_per_step'])"
