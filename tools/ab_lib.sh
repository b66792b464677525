#!/bin/bash
# same-box A/B of two builds of the library: tools/ab_lib.sh <other .so> <bench args...>   (alternating runs, 3 each)
other=$1; shift
for rep in 1 2 3; do
  for v in new old; do
    if [ $v = old ]; then export DVAE_LIB=$other; else unset DVAE_LIB; fi
    timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-profile --no-core-ab 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$v', round(r['ms_per_step'],3))" || exit 1
  done
done
