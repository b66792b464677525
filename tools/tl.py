"""Dump one step of a kernel trace: start(us) dur(us) queue grid kernel.  usage: tl.py <tag> [t_from_us] [t_to_us]"""
import csv, glob, re, sys
tag = sys.argv[1]
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e12
f = glob.glob('gpurun_out/prof_%s/*/*kernel_trace.csv' % tag)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
starts = [i for i, r in enumerate(rows) if 'sched_levels' in r['Kernel_Name']]
a, b = starts[-3], starts[-2]
t0 = int(rows[a]['Start_Timestamp'])
def short(n):
    n = n.replace('void ', '').replace('(anonymous namespace)::', '')
    n = re.sub(r'\(.*', '', n)
    return n[:48]
print('step length %.1f us' % ((int(rows[b]['Start_Timestamp']) - t0) / 1e3))
for r in rows[a:b]:
    s = (int(r['Start_Timestamp']) - t0) / 1e3; e = (int(r['End_Timestamp']) - t0) / 1e3
    if lo <= s <= hi:
        print('%9.1f %8.1f q%s g=%sx%s %s' % (s, e - s, r['Queue_Id'], int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Grid_Size_Y'], short(r['Kernel_Name'])))
