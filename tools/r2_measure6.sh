#!/bin/bash
# round-2 batch 6: gather-loader fused GRU level kernel -- parity (variants + poison), then same-box A/B
O=gpurun_out/r2_10; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_variants.py -x -q --timeout 300 --timeout-method thread -k "fused_level" > $O/var.log 2>&1 || { tail -40 $O/var.log; exit 1; }
tail -1 $O/var.log
DVAE_FUSED_GRU=2 DVAE_POISON_WS=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_large.py -x -q --timeout 300 --timeout-method thread > $O/gather_parity.log 2>&1 || { tail -40 $O/gather_parity.log; exit 1; }
tail -1 $O/gather_parity.log
for B in 4096 512; do
  for v in default gather fused; do
    case $v in default) E="";; gather) E="DVAE_FUSED_GRU=2";; fused) E="DVAE_FUSED_GRU=1";; esac
    env $E timeout -k 10 200 python bench.py --global-batch $B --steps 40 --no-cpu-baseline --no-profile > $O/b_${B}_$v.json 2> $O/b_${B}_$v.err || { echo "bench $B $v failed"; tail -5 $O/b_${B}_$v.err; exit 1; }
    python -c "
import json
r=json.loads(open('$O/b_${B}_$v.json').read().strip().splitlines()[-1])
print('B=$B $v', round(r['ms_per_step'],3),'ms')"
  done
done
# bidirectional ragged config 4: two node sweeps, more levels
for v in default gather; do
  case $v in default) E="";; gather) E="DVAE_FUSED_GRU=2";; esac
  env $E timeout -k 10 300 python bench.py --config c4 --no-cpu-baseline --no-profile > $O/c4_$v.json 2> $O/c4_$v.err || { echo "c4 $v failed"; exit 1; }
  python -c "
import json
r=json.loads(open('$O/c4_$v.json').read().strip().splitlines()[-1])
print('c4 $v', round(r['ms_per_step'],3),'ms')"
done
