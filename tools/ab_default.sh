#!/bin/bash
# usage: tools/ab_default.sh "ENV_1" "ENV_2" ... -- the DEFAULT bench.py run (per-launch events on every 8th step, fp32-core re-run) under each
# environment, alternating, 3 rounds: what the driver's own run measures
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for e in "$@"; do
    v=$(env $e timeout -k 10 300 python bench.py --no-cpu-baseline | python -c "import sys,json; r=json.loads(sys.stdin.readline()); print('%.3f ms  exec %.1f  frac %.3f' % (r['ms_per_step'], r['roofline']['executed_tflops'], r['roofline']['frac']))") || exit 1
    echo "[$e] $v"
  done
done
