#!/bin/bash
# One measurement script for the GPU box (run through gpurun; logs under gpurun_out/<tag>/).  Replaces the per-experiment
# batches of round 2 (tools/r2_measure*.sh): every batch was some mix of these steps.
#   tools/measure.sh configs <tag>                  bench line of every BASELINE configuration (+ c5 stochastic / evaluation path) -> configs.jsonl
#   tools/measure.sh shards <tag>                   the per-GPU shards of the strong-scaled headline batch (B = 2048 / 1024 / 512) -> shards.jsonl
#   tools/measure.sh profile <tag> [bench args]     rocprofv3 --kernel-trace --stats of a short run + per-kernel / per-queue summary
#   tools/measure.sh pmc <tag> [bench args]         the three separate counter passes (MFMA busy + LDS | FETCH_SIZE | WRITE_SIZE) + summary
#   tools/measure.sh ab <tag> "ENV_A" "ENV_B" [bench args]   same-box A/B of two environments, three alternating rounds
#   tools/measure.sh final <tag>                    headline line, profile, counter passes, configs, shards: the end-of-round artefacts
set -o pipefail
cmd=$1; tag=${2:-r03}; shift 2 2>/dev/null
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/$tag; mkdir -p "$O"
line() { python -c "
import json,sys
r=json.loads(sys.stdin.read().strip().splitlines()[-1]); rf=r.get('roofline',{})
print(r['config'].get('name'), r['config'].get('decode',''), 'B=%d' % r['config']['global_batch'], round(r['ms_per_step'],3),'ms', round(r['value']),'DAGs/s', 'frac', round(rf.get('frac',0),3), 'exec', round(rf.get('executed_tflops',0),1), 'cpu', round(r.get('cpu_baseline',{}).get('value',0),2))"; }
case $cmd in
configs)
  : > $O/configs.jsonl
  for a in "--config headline" "--config c1" "--config c2" "--config c3" "--config c4" "--config c5" "--config c5 --stochastic" "--config c5gen" "--config c5gen --stochastic"; do
    timeout -k 10 400 python bench.py $a 2> $O/err.log | tail -1 >> $O/configs.jsonl || { echo "bench $a failed"; tail -5 $O/err.log; exit 1; }
    tail -1 $O/configs.jsonl | line
  done ;;
shards)
  : > $O/shards.jsonl
  for B in 2048 1024 512; do
    timeout -k 10 200 python bench.py --global-batch $B --steps 60 --no-cpu-baseline 2> $O/err.log | tail -1 >> $O/shards.jsonl || { echo "bench B=$B failed"; tail -5 $O/err.log; exit 1; }
    tail -1 $O/shards.jsonl | line
  done ;;
profile)
  cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
  rm -rf gpurun_out/prof_$tag
  timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-core-ab "$@" > $O/prof.log 2>&1 || { tail -5 $O/prof.log; exit 1; }
  python tools/kernel_summary.py gpurun_out/prof_$tag 4 > $O/kernel_summary.txt && python tools/conc.py $tag >> $O/kernel_summary.txt 2>&1
  head -24 $O/kernel_summary.txt; tail -4 $O/kernel_summary.txt ;;
pmc)
  bash tools/pmc2.sh $tag "" --steps 2 --warmup 1 "$@" || exit 1
  python tools/pmc_summ.py $tag 3 > $O/pmc_summary.txt && head -12 $O/pmc_summary.txt ;;
ab)
  A="$1"; B="$2"; shift 2
  bash tools/ab.sh "$A" "$B" "$@" | tee $O/ab.txt ;;
final)
  timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
  cut -c1-300 $O/bench.json
  bash tools/measure.sh profile $tag && bash tools/measure.sh pmc $tag && bash tools/measure.sh configs $tag && bash tools/measure.sh shards $tag ;;
*) echo "usage: tools/measure.sh configs|shards|profile|pmc|ab|final <tag> [...]"; exit 2 ;;
esac
