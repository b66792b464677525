#!/bin/bash
# usage: tools/abn.sh "ENV_1" "ENV_2" ... -- [bench args] -- alternate any number of environments on the same box, 3 rounds
cd $GRAFT_REPO_ROOT
envs=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do envs+=("$1"); shift; done
shift
for r in 1 2 3; do
  for e in "${envs[@]}"; do
    v=$(env $e timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile "$@" | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
    echo "[$e] $v ms"
  done
done
