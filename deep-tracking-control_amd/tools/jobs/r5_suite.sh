# round-end rehearsal on the last commit: whole -m gpu suite, smoke(), the driver's bench command (no profile set)
O=gpurun_out/r5s
mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4 ) > $O/pytest.log 2>&1
tail -6 $O/pytest.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
( time python bench.py --gpus 1 --steps 20 --warmup 5 > $O/driver_like.json 2> $O/driver_like.err ) 2>&1 | grep real
python -c "import json; d=json.load(open('$O/driver_like.json')); print('driver command:', round(d['ms_per_step'],2), 'ms', round(d['value']), 'env-steps/s; frac', round(d['roofline']['frac'],4), 'traffic x', round(d['roofline']['traffic_over_algorithmic'],3), 'cpu', round(d['cpu_baseline']['value']))"
rm -rf gpurun_out/traffic_pmc gpurun_out/gemm_pmc; find gpurun_out -type f -size +4M -delete
