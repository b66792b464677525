#!/bin/bash
# usage: tools/gpu_check.sh <tag> -- GPU tests + bench (on the GPU box), logs under gpurun_out/
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > gpurun_out/test_$1.log 2>&1
rc=$?
tail -3 gpurun_out/test_$1.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_$1.log 2>&1 || exit $?
cut -c1-420 gpurun_out/bench_$1.log | tail -1
