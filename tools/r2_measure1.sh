#!/bin/bash
# round-2 measurement batch 1: new golden tests, per-config bench lines, small-shard host/GPU split
set -o pipefail
O=gpurun_out/r2_04; mkdir -p $O
timeout -k 10 300 python -u -m pytest tests/test_gpu_golden.py -x -q --timeout 200 --timeout-method thread > $O/golden.log 2>&1 || { tail -20 $O/golden.log; exit 1; }
tail -1 $O/golden.log
for c in c1 c2 c3 c4 c5; do
  timeout -k 10 400 python bench.py --config $c > $O/bench_$c.json 2> $O/bench_$c.err || { echo "bench $c failed"; tail -5 $O/bench_$c.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("$O/bench_$c.json").read().strip().splitlines()[-1])
print("$c", round(r["ms_per_step"],3),"ms", round(r["value"]),"DAGs/s frac", round(r["roofline"]["frac"],3), "exec", round(r["roofline"]["executed_tflops"],1), "cpu", round(r["cpu_baseline"]["value"],2))
PY
done
timeout -k 10 200 python tools/host_time.py 512 > $O/host_512.log 2>&1; head -3 $O/host_512.log
timeout -k 10 200 python tools/host_time.py 1024 > $O/host_1024.log 2>&1; head -1 $O/host_1024.log
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_b512 -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --global-batch 512 > $O/prof_b512.log 2>&1
python tools/conc.py b512 > $O/conc_b512.txt 2>&1; cat $O/conc_b512.txt
