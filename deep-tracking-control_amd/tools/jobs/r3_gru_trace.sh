# kernel trace of the gru workload: durations of the per-time-step kernels and the gaps between consecutive kernels of a stream
O=gpurun_out/r3k
mkdir -p $O
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf $R/$O/gru_trace
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$O/gru_trace -o gru --output-format csv -- python $R/bench.py --workload gru --steps 2 --warmup 1 --no-cpu-baseline --no-traffic > /dev/null 2> $R/$O/gru_trace.err
cd $R
python - <<'PY'
import csv, glob, collections
f = glob.glob('gpurun_out/r3k/gru_trace/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').split('(')[0][-40:]
    dur[n].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for n, d in sorted(dur.items(), key=lambda x: -sum(x[1]))[:12]:
    print('%-42s n=%5d avg %7.1f us  total %7.1f ms' % (n, len(d), sum(d) / len(d) / 1e3, sum(d) / 1e6))
# gaps per queue
byq = collections.defaultdict(list)
for r in rows:
    byq[r.get('Queue_Id', '0')].append(r)
for q, rs in byq.items():
    gaps = [int(b['Start_Timestamp']) - int(a['End_Timestamp']) for a, b in zip(rs, rs[1:])]
    gaps = [g for g in gaps if 0 <= g < 200000]
    if len(gaps) > 100:
        gaps.sort()
        print('queue', q, 'kernels', len(rs), 'gap median %.1f us mean %.1f us total %.1f ms' % (gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3, sum(gaps) / 1e6))
t0, t1 = int(rows[0]['Start_Timestamp']), max(int(r['End_Timestamp']) for r in rows)
print('trace span %.1f ms, kernels %d' % ((t1 - t0) / 1e6, len(rows)))
PY
rm -rf $O/gru_trace
