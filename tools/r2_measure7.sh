#!/bin/bash
# round-2 batch 7: windowed forward pair scorer -- full suite, then c3 / c4 / c2 timing
O=gpurun_out/r2_11; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all.log 2>&1 || { tail -40 $O/all.log; exit 1; }
tail -1 $O/all.log
DVAE_POISON_WS=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > $O/poison.log 2>&1 || { tail -40 $O/poison.log; exit 1; }
tail -1 $O/poison.log
for c in c3 c4 c2; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-profile > $O/bench_$c.json 2> $O/bench_$c.err || { echo "bench $c failed"; tail -5 $O/bench_$c.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/bench_$c.json').read().strip().splitlines()[-1])
print('$c', round(r['ms_per_step'],3),'ms', round(r['value']),'DAGs/s')"
done
