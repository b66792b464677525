#!/bin/bash
# usage: tools/ab_batch.sh -- step time at the per-GPU shards of the global 4096 batch (one GPU)
cd $GRAFT_REPO_ROOT
for b in 4096 2048 1024 512; do
  v=$(timeout -k 10 300 python bench.py --no-cpu-baseline --no-profile --global-batch $b --steps 30 | python -c "import sys,json; print('%.3f' % json.loads(sys.stdin.readline())['ms_per_step'])") || exit 1
  echo "B=$b $v ms"
done
