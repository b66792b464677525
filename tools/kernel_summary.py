"""Per-kernel totals of the last two profiled steps of a rocprofv3 kernel trace.
usage: kernel_summary.py <gpurun_out/prof_TAG dir> [steps-to-average]"""
import collections, csv, glob, re, sys
d = sys.argv[1]; nst = int(sys.argv[2]) if len(sys.argv) > 2 else 2
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'sched_levels' in r['Kernel_Name']]
a, b = starts[-2 - nst], starts[-2]
agg = collections.defaultdict(lambda: [0, 0])
for r in rows[a:b]:
    n = re.sub(r'void |\(anonymous namespace\)::|\(.*$', '', r['Kernel_Name'])[:76]
    k = (n, r['Queue_Id'])
    agg[k][0] += 1; agg[k][1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
wall = (int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / nst / 1e6
tot = sum(v[1] for v in agg.values()) / nst / 1e6
print('# %d steps averaged; wall %.2f ms/step under the profiler, kernel time %.2f ms/step, %d launches/step'
      % (nst, wall, tot, sum(v[0] for v in agg.values()) // nst))
qs = collections.defaultdict(lambda: [0, 0])
for (n, q), v in agg.items():
    qs[q][0] += v[0]; qs[q][1] += v[1]
print('# per queue (launches, busy ms per step):', {q: (v[0] // nst, round(v[1] / nst / 1e6, 2)) for q, v in qs.items()})
for (n, q), v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
    print('%-78s q%s n=%4d avg %7.1f us  %8.1f us/step' % (n, q, v[0] // nst, v[1] / v[0] / 1e3, v[1] / nst / 1e3))
# gaps between consecutive launches of the busiest queue (the per-level chain): idle time a launch waits for
busiest = max(qs, key=lambda q: qs[q][1])
ch = [r for r in rows[a:b] if r['Queue_Id'] == busiest]
gaps = [int(ch[i + 1]['Start_Timestamp']) - int(ch[i]['End_Timestamp']) for i in range(len(ch) - 1)]
gaps = [g for g in gaps if g < 200000]   # step boundaries (host work) aside
pos = [g for g in gaps if g > 0]
print('# queue %s: %d launches/step, gaps > 0: %d/step, mean %.2f us, total %.2f ms/step' % (
    busiest, len(ch) // nst, len(pos) // nst, sum(pos) / max(1, len(pos)) / 1e3, sum(pos) / nst / 1e6))
