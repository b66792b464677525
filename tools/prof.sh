#!/bin/bash
# usage: tools_prof.sh <tag>  -- kernel-trace profile of a short bench run (on the GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_$1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$1 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline $PROF_ARGS > gpurun_out/prof_$1.log 2>&1
tail -1 gpurun_out/prof_$1.log | cut -c1-200
