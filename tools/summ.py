import csv, collections, glob, sys
tag = sys.argv[1]
d = glob.glob('gpurun_out/prof_%s/*/*kernel_trace.csv' % tag)[0]
rows = list(csv.DictReader(open(d)))
agg = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name']
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'gemm_f32_kernel' in n:
        v = n.split('gemm_f32_kernel')[1].split('(')[0]
        key = ('gemm' + v, int(r['Grid_Size_X']) // 256, r['Grid_Size_Y'])
    else:
        key = (n.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:60],)
    agg[key].append(dur)
tot = sum(sum(v) for v in agg.values())
steps = 13.0
print('total %.2f ms/step' % (tot / 1e3 / steps))
for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print('%-75s n=%5d avg %8.1f us  %7.3f ms/step' % (str(k), len(v), sum(v) / len(v), sum(v) / 1e3 / steps))
