#!/bin/bash
# kernel trace of config 4 (ragged 8..128 nodes, bidirectional, B=512) and config 2
O=gpurun_out/r2_20; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in c4; do
  rm -rf gpurun_out/prof_$c
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_$c -- python3 bench.py --config $c --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-core-ab > $O/prof_$c.log 2>&1 || { tail -5 $O/prof_$c.log; exit 1; }
  python tools/kernel_summary.py gpurun_out/prof_$c 2 > $O/trace_$c.txt || exit 1
  head -32 $O/trace_$c.txt; tail -1 $O/trace_$c.txt
  rm -rf gpurun_out/prof_$c
done
