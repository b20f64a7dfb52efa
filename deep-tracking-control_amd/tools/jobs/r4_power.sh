# average socket power and shader clock while the bench step runs (rocm-smi polled beside a long bench run), and while the bare MFMA
# stream runs on zero / random operands
O=gpurun_out/r4p
mkdir -p $O
( which rocm-smi amd-smi; rocm-smi --showpower --showclocks 2>&1 | head -30 ) > $O/r04_power.txt 2>&1
sample() {   # $1 = label, runs until the background job $2 ends
  n=0
  while kill -0 $2 2>/dev/null; do
    rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk" | tr '\n' ' ' | sed "s/^/$1: /" >> $O/r04_power.txt; echo >> $O/r04_power.txt
    n=$((n+1)); sleep 0.5
  done
}
python bench.py --steps 150 --warmup 5 --no-cpu-baseline --no-traffic > $O/power_bench.json 2>/dev/null &
P=$!; sleep 8; sample "bench step" $P
python - > $O/power_probe.log 2>&1 <<'PY' &
import sys, time
sys.path.insert(0, 'deep-tracking-control_amd')
from dtc_amd import ops
t0 = time.time()
while time.time() - t0 < 12: ops.mfma_sustained('cuda:0', False, launches=40)
print('zero done'); sys.stdout.flush()
t0 = time.time()
while time.time() - t0 < 12: ops.mfma_sustained('cuda:0', True, launches=40)
PY
P=$!; sleep 6; sample "mfma stream (zero operands first ~12 s, then random)" $P
grep -v amdgpu $O/r04_power.txt | tail -70
