#!/bin/bash
# round-2 bench lines of every BASELINE configuration + the per-GPU shards of the strong-scaled headline batch
O=gpurun_out/r2_13; mkdir -p $O; : > $O/configs.jsonl
for c in headline c1 c2 c3 c4 c5; do
  timeout -k 10 400 python bench.py --config $c 2> $O/err_$c.log | tail -1 >> $O/configs.jsonl || { echo "bench $c failed"; tail -5 $O/err_$c.log; exit 1; }
  tail -1 $O/configs.jsonl | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print(r['config']['name'], round(r['ms_per_step'],3),'ms', round(r['value']),'DAGs/s frac', round(r['roofline']['frac'],3), 'exec', round(r['roofline']['executed_tflops'],1), 'whole', round(r['roofline']['whole_step_tflops'],1), 'cpu', round(r['cpu_baseline']['value'],2))"
done
: > $O/shards.jsonl
for B in 2048 1024 512; do
  timeout -k 10 200 python bench.py --global-batch $B --steps 60 --no-cpu-baseline 2> $O/err_b$B.log | tail -1 >> $O/shards.jsonl || { echo "bench B=$B failed"; exit 1; }
  tail -1 $O/shards.jsonl | python -c "
import json,sys
r=json.loads(sys.stdin.read())
print('B=$B', round(r['ms_per_step'],3),'ms', round(r['value']),'DAGs/s')"
done
