#!/bin/bash
O=gpurun_out/r2_18; mkdir -p $O
run() {
  local label=$1 E=$2; shift 2
  env $E timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-profile > $O/b.json 2> $O/b.err || { echo "bench $label failed"; tail -5 $O/b.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$label', round(r['ms_per_step'],3),'ms')"
}
for rep in 1 2; do
  for v in 1 0; do
    run "c1 x6=$v" DVAE_GEMM_X6=$v --config c1 --steps 100 || exit 1
    run "B512 x6=$v" DVAE_GEMM_X6=$v --global-batch 512 --steps 50 || exit 1
    run "B1024 x6=$v" DVAE_GEMM_X6=$v --global-batch 1024 --steps 50 || exit 1
    run "c4 x6=$v" DVAE_GEMM_X6=$v --config c4 || exit 1
  done
done
