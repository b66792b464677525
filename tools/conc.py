"""Concurrency summary of one step of a kernel trace. usage: conc.py <tag> [scheds_per_step]"""
import csv, glob, sys
tag = sys.argv[1]; sps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob('gpurun_out/prof_%s/*/*kernel_trace.csv' % tag)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'sched_levels' in r['Kernel_Name']]
a, b = starts[-1 - 3 * sps], starts[-1 - sps]
t0 = int(rows[a]['Start_Timestamp'])
print('ms/step %.2f' % ((int(rows[b]['Start_Timestamp']) - t0) / 2e6))
ev = []
for r in rows[a:b]:
    g = 'gemm_f32' in r['Kernel_Name']
    ev.append((int(r['Start_Timestamp']), 1, g)); ev.append((int(r['End_Timestamp']), -1, g))
ev.sort()
n = ng = 0; last = None; hist = {}; gh = {}
for t, d, g in ev:
    if last is not None:
        hist[n] = hist.get(n, 0) + t - last
        gh[ng] = gh.get(ng, 0) + t - last
    n += d; ng += d if g else 0; last = t
tot = sum(hist.values())
print('kernels running:', {k: round(v / tot, 3) for k, v in sorted(hist.items())})
print('gemm kernels running:', {k: round(v / tot, 3) for k, v in sorted(gh.items())})
qs = {}
for r in rows[a:b]:
    q = qs.setdefault(r['Queue_Id'], [0, 0]); q[0] += 1; q[1] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
print('per queue (launches, busy ms per step):', {k: (v[0] // 2, round(v[1] / 2e6, 2)) for k, v in qs.items()})
