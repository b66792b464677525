#!/bin/bash
# round-2 batch 3: side-work deferral and split Adam -- parity, then same-box A/B at B = 512 / 1024 / 2048 / 4096
O=gpurun_out/r2_06; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all.log 2>&1 || { tail -30 $O/all.log; exit 1; }
tail -1 $O/all.log
for B in 512 1024 2048 4096; do
  for v in new nodefer oneopt old; do
    case $v in new) E=""; A="";; nodefer) E="DVAE_DEFER_SIDE_ROWS=0"; A="";; oneopt) E=""; A="--single-optimizer";; old) E="DVAE_DEFER_SIDE_ROWS=0"; A="--single-optimizer";; esac
    env $E timeout -k 10 200 python bench.py --global-batch $B --steps 40 --no-cpu-baseline --no-profile $A > $O/b_${B}_$v.json 2> $O/b_${B}_$v.err || { echo "bench $B $v failed"; tail -5 $O/b_${B}_$v.err; exit 1; }
    python -c "
import json
r=json.loads(open('$O/b_${B}_$v.json').read().strip().splitlines()[-1])
print('B=$B $v', round(r['ms_per_step'],3),'ms')"
  done
done
