#!/bin/bash
# round 6: first GPU pass of the persistent GRU forward + the pending test changes
cd "$(dirname "$0")/../../.." || exit 1
out=gpurun_out/r06_seq1.txt
: > $out
echo "== test_hip_gru" >> $out
timeout 900 python -m pytest tests/test_hip_gru.py -x -q -m gpu -s 2>&1 | grep -v "amdgpu.ids" | tail -25 >> $out
echo "== gru / composite path" >> $out
timeout 1500 python -m pytest tests/test_gru_path.py tests/test_composite_path.py tests/test_lstm_path.py -x -q -m gpu 2>&1 | tail -8 >> $out
echo "== dp" >> $out
timeout 1500 python -m pytest tests/test_hip_dp.py tests/test_hip_dp_g7.py -x -q -m gpu 2>&1 | tail -8 >> $out
for w in gru composite; do
  for seq in 1 0; do
    echo "== bench --workload $w DTC_GRU_SEQ=$seq" >> $out
    DTC_GRU_SEQ=$seq timeout 600 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-traffic --no-in-situ 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print({k: d.get(k) for k in ('value','ms_per_step')}, d.get('roofline',{}).get('frac'))" >> $out 2>&1
  done
done
cat $out
