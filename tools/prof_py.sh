#!/bin/bash
# usage: tools/prof_py.sh <tag> <script> [args]: kernel-trace profile of a python script (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
rm -rf gpurun_out/prof_$tag
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 "$@" > gpurun_out/prof_$tag.log 2>&1
tail -2 gpurun_out/prof_$tag.log | cut -c1-200
