#!/bin/bash
# round-2 batch 4: fused level kernels at small batches (env knobs only), same box
O=gpurun_out/r2_07; mkdir -p $O
for B in 32 512 1024; do
  for v in default gm0 gm0_gru; do
    case $v in default) E="";; gm0) E="DVAE_FUSED_GM_MIN=0";; gm0_gru) E="DVAE_FUSED_GM_MIN=0 DVAE_FUSED_GRU=1";; esac
    env $E timeout -k 10 200 python bench.py --global-batch $B --steps 60 --no-cpu-baseline --no-profile > $O/b_${B}_$v.json 2> $O/b_${B}_$v.err || { echo "bench $B $v failed"; tail -5 $O/b_${B}_$v.err; exit 1; }
    python -c "
import json
r=json.loads(open('$O/b_${B}_$v.json').read().strip().splitlines()[-1])
print('B=$B $v', round(r['ms_per_step'],3),'ms')"
  done
done
