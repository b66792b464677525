#!/bin/bash
# bf16x6 GEMM core: whole GPU suite under it, then same-box A/B of every configuration
O=gpurun_out/r2_15; mkdir -p $O
DVAE_GEMM_X6=1 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all_x6.log 2>&1 || { tail -40 $O/all_x6.log; exit 1; }
tail -1 $O/all_x6.log
run() {
  local label=$1 E=$2; shift 2
  env $E timeout -k 10 300 python bench.py "$@" --no-cpu-baseline > $O/b.json 2> $O/b.err || { echo "bench $label failed"; tail -5 $O/b.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$label', round(r['ms_per_step'],3),'ms', r['roofline'].get('achieved'), r['roofline'].get('executed_tflops'))"
}
for v in 1 0; do
  run "headline x6=$v" DVAE_GEMM_X6=$v || exit 1
  run "c2 x6=$v" DVAE_GEMM_X6=$v --config c2 || exit 1
  run "c3 x6=$v" DVAE_GEMM_X6=$v --config c3 || exit 1
  run "c4 x6=$v" DVAE_GEMM_X6=$v --config c4 || exit 1
  run "c5 x6=$v" DVAE_GEMM_X6=$v --config c5 || exit 1
  run "B512 x6=$v" DVAE_GEMM_X6=$v --global-batch 512 --steps 50 || exit 1
done
