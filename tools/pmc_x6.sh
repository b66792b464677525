#!/bin/bash
# counter passes over one GEMM shape per core: MFMA busy, VALU / LDS activity, bank conflicts
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_x6; rm -rf $O; mkdir -p $O
for core in 1 0; do
 for pass in "a:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU" "b:SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU"; do
  name=${pass%%:*}; ctr=${pass#*:}
  timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d $O/c${core}_$name -- python3 tools/x6_one.py "$@" $core > $O/c${core}_$name.log 2>&1 || { echo "pass $name failed"; tail -3 $O/c${core}_$name.log; exit 1; }
 done
done
python - <<'PY'
import csv, glob, collections
for core in (1, 0):
    tot = collections.defaultdict(float); n = 0
    for ps in 'ab':
        for f in glob.glob('gpurun_out/pmc_x6/c%d_%s/*/*counter_collection.csv' % (core, ps)):
            for r in csv.DictReader(open(f)):
                if 'gemm_f32_kernel' in r['Kernel_Name']:
                    tot[r['Counter_Name']] += float(r['Counter_Value'])
    print('core', core, {k: '%.3g' % v for k, v in sorted(tot.items())})
PY
