#!/bin/bash
# LDS-sliced pair forward for graphs of more than 16 nodes: parity, then same-box A/B at c3 / c4
O=gpurun_out/r2_24; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests/test_gpu_large.py tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_gpu_golden.py -x -q --timeout 200 --timeout-method thread -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
for c in c3 c4; do
  for v in slice old slice old; do
    case $v in slice) E="A=1";; old) E="DVAE_NO_PAIR_FWD_SLICE=1";; esac
    env $E timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-profile --no-core-ab > $O/b.json 2> $O/b.err || { echo "bench $c failed"; tail -5 $O/b.err; exit 1; }
    python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c $v', round(r['ms_per_step'],3),'ms')"
  done
done
