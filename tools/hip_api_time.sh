#!/bin/bash
# host-side cost of the HIP API calls of a small-batch step: rocprofv3 --hip-runtime-trace --stats (no counters)
B=${1:-32}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/hipapi_$B
timeout -k 10 300 rocprofv3 --hip-runtime-trace --stats --output-format csv -d gpurun_out/hipapi_$B -- python3 bench.py --global-batch $B --steps 50 --warmup 5 --no-cpu-baseline --no-profile --no-core-ab > gpurun_out/hipapi_$B.log 2>&1 || { tail -5 gpurun_out/hipapi_$B.log; exit 1; }
f=$(ls gpurun_out/hipapi_$B/*/*hip_api_stats.csv 2>/dev/null | head -1)
[ -z "$f" ] && { ls -R gpurun_out/hipapi_$B | head; exit 1; }
head -14 $f | cut -c1-160
tail -1 gpurun_out/hipapi_$B.log | cut -c1-200
rm -rf gpurun_out/hipapi_$B
