#!/bin/bash
# lin01 chain as one launch per step: parity, then c2 / c4
O=gpurun_out/r2_21; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_large.py tests/test_gpu_variants.py tests/test_gpu_fullsize.py -x -q --timeout 200 --timeout-method thread -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for c in c2 c4 c2 c4; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-profile --no-core-ab > $O/b.json 2> $O/b.err || { echo "bench $c failed"; tail -5 $O/b.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c', round(r['ms_per_step'],3),'ms')"
done
