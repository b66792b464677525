#!/bin/bash
# fused level kernels on the x6 core: parity (variants test drives every fused form), then same-box A/B
O=gpurun_out/r2_19; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_variants.py tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -1 $O/tests.log
run() {
  local label=$1 E=$2; shift 2
  env $E timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-profile --no-core-ab > $O/b.json 2> $O/b.err || { echo "bench $label failed"; tail -5 $O/b.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$label', round(r['ms_per_step'],3),'ms')"
}
for rep in 1 2; do
  run "headline unfused" "A=1" || exit 1
  run "headline fused-gm" "DVAE_FUSED_GM_MIN=2048" || exit 1
  run "headline fused-gm+gru" "DVAE_FUSED_GM_MIN=2048 DVAE_FUSED_GRU=1" || exit 1
  run "headline fused-gru" "DVAE_FUSED_GRU=1" || exit 1
  run "c3 unfused" "A=1" --config c3 || exit 1
  run "c3 fused-gm" "DVAE_FUSED_GM_MIN=2048" --config c3 || exit 1
  run "c3 fused-gm+gru" "DVAE_FUSED_GM_MIN=2048 DVAE_FUSED_GRU=1" --config c3 || exit 1
  run "c2 unfused" "A=1" --config c2 || exit 1
  run "c2 fused-gru" "DVAE_FUSED_GRU=1" --config c2 || exit 1
done
