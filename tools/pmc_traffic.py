"""profiles/r03_pmc_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of tools/pmc2.sh (run where gpurun_out/ is):
    python tools/pmc_traffic.py <config name> <pmc tag> <steps in the profiled run>
Per config: the GEMM family's HBM bytes per launch (bench.py's roofline.traffic) and the per-step bytes of the HBM-bound
kernel families bench.py times (roofline.hbm[*].traffic_GB_per_step).  FETCH_SIZE is doubled (gfx950 reports half of a
wide coalesced read, MI355X_MICROARCH.md) and both counters are in KiB."""
import collections, csv, glob, json, os, sys
cfg, tag, steps = sys.argv[1], sys.argv[2], float(sys.argv[3])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name, counter):
    # (gpurun merges each call's files into gpurun_out/ without removing older ones: take the newest pass)
    f = max(glob.glob(os.path.join(ROOT, 'gpurun_out/pmc_%s_%s/*/*counter_collection.csv' % (tag, name))), key=os.path.getmtime)
    by, calls, seen = collections.defaultdict(float), collections.Counter(), set()
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0]
        by[k] += float(r['Counter_Value']) * 1024.0
        if r['Dispatch_Id'] not in seen:
            seen.add(r['Dispatch_Id']); calls[k] += 1
    return by, calls


fe, calls = load('fetch', 'FETCH_SIZE')
wr, _ = load('write', 'WRITE_SIZE')
total = lambda pred: sum(2.0 * fe[k] + wr.get(k, 0.0) for k in fe if pred(k))
ncalls = lambda pred: sum(calls[k] for k in calls if pred(k))
is_gemm = lambda k: k.startswith('gemm_f32_kernel') or k.startswith('level_x6_kernel') or k.startswith('level_kernel') or k.startswith('small_kernel')
fam = {"0": lambda k: k.startswith('pair_fwd'), "1": lambda k: k.startswith('pair_bwd'),
       "2": lambda k: k.startswith('agg_fwd') or k.startswith('prefix_gru_fwd'),
       "3": lambda k: k.startswith('dgm_bwd') or k.startswith('dgm_gather')}
entry = {"gemm_bytes_per_launch": total(is_gemm) / max(ncalls(is_gemm), 1), "gemm_launches_per_step": ncalls(is_gemm) / steps,
         "gemm_GB_per_step": total(is_gemm) / steps / 1e9,
         "hbm": {k: total(p) / steps / 1e9 for k, p in fam.items()}}
path = os.path.join(ROOT, 'profiles', 'r03_pmc_traffic.json')
d = json.load(open(path)) if os.path.exists(path) else {}
d["_how"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/pmc2.sh), FETCH_SIZE x 2 per the gfx950 note, KiB -> bytes; tools/pmc_traffic.py"
d[cfg] = entry
json.dump(d, open(path, 'w'), indent=1, sort_keys=True)
print(json.dumps(entry, indent=1))
