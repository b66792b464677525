#!/bin/bash
# round-2 batch 2: windowed pair backward -- parity, then c2/c3/c4 timing with window 16 / 8 / two-kernel form
O=gpurun_out/r2_05; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all.log 2>&1 || { tail -30 $O/all.log; exit 1; }
tail -1 $O/all.log
DVAE_POISON_WS=1 timeout -k 10 300 python -u -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py -x -q --timeout 300 --timeout-method thread > $O/poison.log 2>&1 || { tail -30 $O/poison.log; exit 1; }
tail -1 $O/poison.log
for c in c3 c2 c4; do
  for v in win16 win8 two; do
    case $v in win16) E="";; win8) E="DVAE_PAIR_WIN=8";; two) E="DVAE_NO_PAIR_WIN=1";; esac
    env $E timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-profile > $O/bench_${c}_$v.json 2> $O/bench_${c}_$v.err || { echo "bench $c $v failed"; tail -5 $O/bench_${c}_$v.err; exit 1; }
    python -c "
import json
r=json.loads(open('$O/bench_${c}_$v.json').read().strip().splitlines()[-1])
print('$c $v', round(r['ms_per_step'],3),'ms', round(r['value']),'DAGs/s')"
  done
done
