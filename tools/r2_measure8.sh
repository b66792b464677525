#!/bin/bash
# same-box A/B of the windowed forward pair scorer + its kernel time from a kernel trace
O=gpurun_out/r2_12; mkdir -p $O
for c in c3 c4; do
  for v in win old win old; do
    case $v in win) E="";; old) E="DVAE_NO_PAIR_FWD_WIN=1";; esac
    env $E timeout -k 10 300 python bench.py --config $c --no-cpu-baseline --no-profile > $O/b.json 2> $O/b.err || { echo "bench $c failed"; tail -5 $O/b.err; exit 1; }
    python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$c $v', round(r['ms_per_step'],3),'ms')"
  done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in win old; do
  case $v in win) E="";; old) E="DVAE_NO_PAIR_FWD_WIN=1";; esac
  rm -rf gpurun_out/prof_c3$v
  env $E timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3$v -- python3 bench.py --config c3 --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $O/prof_$v.log 2>&1
  grep -h "pair_" gpurun_out/prof_c3$v/*/*kernel_stats.csv | cut -c1-60,200-400 | head -4
done
