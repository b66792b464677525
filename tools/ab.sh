#!/bin/bash
# usage: tools/ab.sh "ENV_A" "ENV_B" [bench args] -- alternate two environments on the same box, 3 rounds
cd $GRAFT_REPO_ROOT
A="$1"; B="$2"; shift 2
for r in 1 2 3; do
  for e in "$A" "$B"; do
    v=$(env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile "$@" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "[$e] $v ms"
  done
done
