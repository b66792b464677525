#!/bin/bash
O=gpurun_out/r2_25; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c3 c4; do
  rm -rf gpurun_out/prof_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --config $c --steps 3 --warmup 1 --no-cpu-baseline --no-profile --no-core-ab > $O/prof_$c.log 2>&1 || { tail -5 $O/prof_$c.log; exit 1; }
  grep -h "pair_" gpurun_out/prof_$c/*/*kernel_stats.csv | cut -c1-50,120-260
  rm -rf gpurun_out/prof_$c
done
