#!/bin/bash
# small-level kernels wired into the sweeps: variant test, then same-box A/B at the small shards, c1, c2, c4 and the headline
O=gpurun_out/r2_13; mkdir -p $O
timeout -k 10 400 python -u -m pytest tests/test_gpu_variants.py tests/test_gpu_kernels.py -x -q --timeout 120 --timeout-method thread -m gpu > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
run() {  # label env args...
  local label=$1 E=$2; shift 2
  env $E timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-profile > $O/b.json 2> $O/b.err || { echo "bench $label failed"; tail -5 $O/b.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$label', round(r['ms_per_step'],3),'ms')"
}
for v in small off small off; do
  case $v in small) E="DVAE_SMALL_ROWS=1024";; off) E="DVAE_SMALL_ROWS=0";; esac
  run "B32 $v" $E --global-batch 32 --steps 100 || exit 1
  run "B512 $v" $E --global-batch 512 --steps 50 || exit 1
  run "B1024 $v" $E --global-batch 1024 --steps 50 || exit 1
done
for v in small off small2k; do
  case $v in small) E="DVAE_SMALL_ROWS=1024";; off) E="DVAE_SMALL_ROWS=0";; small2k) E="DVAE_SMALL_ROWS=2048";; esac
  run "c2 $v" $E --config c2 || exit 1
  run "c4 $v" $E --config c4 || exit 1
  run "headline $v" $E || exit 1
  run "B2048 $v" $E --global-batch 2048 --steps 30 || exit 1
done
