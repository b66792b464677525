#!/bin/bash
# round-2 batch 5: C-level train step -- full GPU suite (+ poisoned workspaces), then timing at B = 32 / 512 / 1024 / 4096
O=gpurun_out/r2_09; mkdir -p $O
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all.log 2>&1 || { tail -40 $O/all.log; exit 1; }
tail -1 $O/all.log
DVAE_POISON_WS=1 timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_dp.py -x -q --timeout 300 --timeout-method thread > $O/poison.log 2>&1 || { tail -40 $O/poison.log; exit 1; }
tail -1 $O/poison.log
for B in 32 512 1024 4096; do
  timeout -k 10 200 python bench.py --global-batch $B --steps 60 --no-cpu-baseline --no-profile > $O/b_$B.json 2> $O/b_$B.err || { echo "bench $B failed"; tail -5 $O/b_$B.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b_$B.json').read().strip().splitlines()[-1])
print('B=$B', round(r['ms_per_step'],3),'ms')"
done
timeout -k 10 200 python tools/host_time.py 512 2>&1 | grep "host enqueue"
