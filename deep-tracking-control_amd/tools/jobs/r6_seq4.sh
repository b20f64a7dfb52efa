#!/bin/bash
cd "$(dirname "$0")/../../.." || exit 1
out=gpurun_out/r06_seq4.txt
: > $out
echo "== test_hip_gru" >> $out
timeout 900 python -m pytest tests/test_hip_gru.py -x -q -m gpu 2>&1 | tail -4 >> $out
echo "== dp repeat test" >> $out
timeout 1200 python -m pytest "tests/test_hip_dp.py::test_bucketed_exchange_on_the_side_stream_equals_one_exchange_after_the_join" -x -q -m gpu 2>&1 | grep -v "socket.cpp\|amdgpu.ids" | grep -E "AssertionError|passed|failed" | cut -c1-400 | head -8 >> $out
echo "== gru / composite path tests" >> $out
timeout 1500 python -m pytest tests/test_gru_path.py tests/test_composite_path.py -x -q -m gpu 2>&1 | tail -6 >> $out
for w in gru composite; do
  for pair in 1 0; do
    echo "== bench --workload $w DTC_GRU_SEQ_PAIR=$pair" >> $out
    DTC_GRU_SEQ_PAIR=$pair timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ms_per_step')}, d.get('roofline',{}).get('frac'))
det=json.load(open('gpurun_out/bench_detail.json'))
for k,v in sorted(det['kernel_classes'].items(), key=lambda kv:-kv[1]['ms'])[:4]: print('   ',k,v)
" >> $out 2>&1
  done
done
cat $out
