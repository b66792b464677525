#!/bin/bash
# usage: tools/ab4.sh "ENV1" "ENV2" ... -- [bench args]: alternate environments on one box, 2 rounds
cd $GRAFT_REPO_ROOT
envs=(); while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs+=("$1"); shift; done; shift
for r in 1 2; do
  for e in "${envs[@]}"; do
    v=$(env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile "$@" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "[$e] $v ms"
  done
done
