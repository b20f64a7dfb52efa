# round-end rehearsal: the whole GPU suite, smoke(), default bench -> gpurun_out/r2/
mkdir -p gpurun_out/r2
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r2/full_tests.log
tail -5 gpurun_out/r2/full_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE OK')" 2>&1 | tail -3
( time timeout 900 python bench.py > gpurun_out/r2/full_bench.json 2> gpurun_out/r2/full_bench.err ) 2>&1 | grep real
python - <<PY
import json
d=json.loads(open('gpurun_out/r2/full_bench.json').read().strip().splitlines()[-1])
print('VALUE', d['value'], 'ms', d['ms_per_step'], 'roof', d['roofline']['achieved'], d['roofline']['frac'], 'traffic', d['roofline']['traffic'], 'cpu', d['cpu_baseline']['value'])
PY
