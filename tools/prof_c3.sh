#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_c3
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -- python3 tools/prof_c3.py $1 > gpurun_out/prof_c3.log 2>&1
tail -2 gpurun_out/prof_c3.log | cut -c1-200
