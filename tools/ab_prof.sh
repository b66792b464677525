#!/bin/bash
# usage: tools/ab_prof.sh -- alternate bench runs with and without per-launch GEMM events (same box)
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do
  for a in "" "--no-profile"; do
    v=$(timeout -k 10 300 python bench.py --no-cpu-baseline $a | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "[$a] $v ms"
  done
done
