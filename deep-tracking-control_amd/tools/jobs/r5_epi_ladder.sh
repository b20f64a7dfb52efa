# (1) timing ladder of linear_h2i_kernel's epilogue: variants built with -DDTC_H2I_PROBE=1..5 (tools/build_variant.sh epi<l> gemm_h2i.hip ...)
#     drop parts of the epilogue from the end (results WRONG, timing only): 1 = no split / image stores, 2 = + no row-maximum exchange,
#     3 = + no per-element maxima, 4 = + no LDS transposition, 5 = + no bias / sign record / activation
# (2) kernel trace of the composite workload's overlapped step: per-kernel durations, per-queue gaps, concurrency
O=gpurun_out/r5e
mkdir -p $O
R=$PWD
T=deep-tracking-control_amd/tools
for rep in 1 2; do
  python $T/h2i_probe.py all time 2>&1 | grep -v Warn | tr '\n' ' '; echo " | product"
  for l in 1 2 3 4 5; do
    DTC_LIB=$R/$T/_bin/libdtc_hip_epi$l.so DTC_SKIP_ABI_CHECK=1 python $T/h2i_probe.py all time 2>&1 | grep -v Warn | tr '\n' ' '; echo " | epi$l"
  done
done | tee $O/ladder.txt
cd /tmp && export TMPDIR=/tmp
for w in composite gru; do
rm -rf $R/$O/${w}_trace
timeout 600 rocprofv3 --kernel-trace -d $R/$O/${w}_trace -o $w --output-format csv -- python $R/bench.py --workload $w --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > /dev/null 2> $R/$O/${w}_trace.err
cd $R
python - $w <<'PY' | tee $O/${w}_trace.txt
import csv, glob, collections, sys
w = sys.argv[1]
f = glob.glob(f'gpurun_out/r5e/{w}_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'foothold_plan' in r['Kernel_Name']]
print('steps', len(starts))
a = starts[-2]; b = starts[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp']); t1 = max(int(r['End_Timestamp']) for r in step)
print('step span %.2f ms, %d dispatches' % ((t1 - t0) / 1e6, len(step)))
dur = collections.defaultdict(list)
for r in step:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-44:]
    dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for n, d in sorted(dur.items(), key=lambda x: -sum(x[1]))[:16]:
    print('%-46s n=%5d avg %7.1f us  total %7.2f ms' % (n, len(d), sum(d) / len(d) / 1e3, sum(d) / 1e6))
byq = collections.defaultdict(list)
for r in step: byq[r.get('Queue_Id', '0')].append(r)
for q, rs in byq.items():
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in rs)
    gaps = [int(y['Start_Timestamp']) - int(x['End_Timestamp']) for x, y in zip(rs, rs[1:])]
    gaps = sorted(g for g in gaps if g >= 0)
    print('queue', q, 'kernels', len(rs), 'busy %.1f ms' % (busy / 1e6), 'gap median %.1f us, sum %.1f ms' % ((gaps[len(gaps) // 2] / 1e3 if gaps else 0), sum(gaps) / 1e6))
# concurrency histogram + time during which a gru step kernel is running, alone or not
pts = []
for r in step:
    g = 'gru' in r['Kernel_Name']
    pts.append((int(r['Start_Timestamp']), 1, g)); pts.append((int(r['End_Timestamp']), -1, g))
pts.sort()
cur = 0; curg = 0; last = t0
hist = collections.Counter(); gh = collections.Counter()
for t, d, g in pts:
    hist[cur] += t - last
    if curg: gh[(curg, cur)] += t - last
    last = t; cur += d
    if g: curg += d
print('concurrency (kernels in flight -> ms):', {k: round(v / 1e6, 2) for k, v in sorted(hist.items())})
print('while gru kernels run ((gru kernels, all kernels) -> ms):', {k: round(v / 1e6, 2) for k, v in sorted(gh.items())})
PY
rm -rf $R/$O/${w}_trace
cd /tmp
done
