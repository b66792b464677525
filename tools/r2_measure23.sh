#!/bin/bash
# DPP wave_sum: parity (whole suite), kernel times of the pair scorers, step times
O=gpurun_out/r2_26; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all.log 2>&1 || { tail -40 $O/all.log; exit 1; }
tail -1 $O/all.log
bash tools/r2_measure22.sh || exit 1
for c in headline c3 c4 c2; do
  timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-profile --no-core-ab > $O/b.json 2> $O/b.err || { echo "bench $c failed"; tail -5 $O/b.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c', round(r['ms_per_step'],3),'ms')"
done
