#!/bin/bash
# round-2 headline profile: bench line + rocprofv3 kernel trace of the same command
O=gpurun_out/r2_08; mkdir -p $O
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
cut -c1-400 $O/bench.json
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof_r02
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-core-ab > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
python tools/kernel_summary.py gpurun_out/prof_r02 4 > $O/kernel_summary.txt; head -30 $O/kernel_summary.txt
python tools/conc.py r02 >> $O/kernel_summary.txt 2>&1; tail -4 $O/kernel_summary.txt
