"""Timeline analysis of a kernel trace: idle time, GEMM-busy time, non-GEMM-only time per step."""
import csv, glob, sys
tag = sys.argv[1]; steps = int(sys.argv[2]) if len(sys.argv) > 2 else 13
rows = list(csv.DictReader(open(glob.glob('gpurun_out/prof_%s/*/*kernel_trace.csv' % tag)[0])))
ev = []
for r in rows:
    g = 'gemm_f32_kernel' in r['Kernel_Name']
    ev.append((int(r['Start_Timestamp']), 1, g)); ev.append((int(r['End_Timestamp']), -1, g))
ev.sort()
# restrict to the last `steps-3` steps region roughly: use whole trace after first 20% of time
starts = sorted(int(r['Start_Timestamp']) for r in rows if 'sched_levels' in r['Kernel_Name'])
t0 = starts[-10]; nsteps = 10
ng = nn = 0; last = None; idle = gemm = other = both = 0
for t, d, g in ev:
    if last is not None and t > t0:
        dt = t - max(last, t0)
        if dt > 0:
            if ng == 0 and nn == 0: idle += dt
            elif ng > 0: gemm += dt
            else: other += dt
    if g: ng += d
    else: nn += d
    last = t
tot = idle + gemm + other
print('%.2f ms/step;' % (tot / 1e6 / nsteps), 'window %.1f ms: idle %.1f%%  gemm-running %.1f%%  only-non-gemm %.1f%%' % (tot / 1e6, 100 * idle / tot, 100 * gemm / tot, 100 * other / tot))
