#!/bin/bash
# usage: tools/ab_dist.sh [bench args] -- RCCL rehearsal (world size 1) with and without schedule prefetch
cd $GRAFT_REPO_ROOT
for r in 1 2; do
  for a in "" "--no-prefetch"; do
    v=$(DVAE_FORCE_DIST=1 timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile $a "$@" 2>/dev/null | grep '^{' | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "[dist $a $*] $v ms"
  done
done
