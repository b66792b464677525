#!/bin/bash
# kernel trace of the B = 512 / B = 32 steps with the small-level kernels
O=gpurun_out/r2_14; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for B in 512 32; do
  rm -rf gpurun_out/prof_b$B
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_b$B -- python3 bench.py --global-batch $B --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-core-ab > $O/prof_$B.log 2>&1 || { tail -5 $O/prof_$B.log; exit 1; }
  python tools/kernel_summary.py gpurun_out/prof_b$B 2 > $O/trace_b$B.txt || exit 1
  head -45 $O/trace_b$B.txt
  rm -rf gpurun_out/prof_b$B
done
