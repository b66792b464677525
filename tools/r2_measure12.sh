#!/bin/bash
# x6 default: fused gate/map level kernel (fp32 MFMA) against GEMM(x6) + epilogue kernel; small-level kernels on/off
O=gpurun_out/r2_16; mkdir -p $O
run() {
  local label=$1 E=$2; shift 2
  env $E timeout -k 10 300 python bench.py "$@" --no-cpu-baseline --no-profile > $O/b.json 2> $O/b.err || { echo "bench $label failed"; tail -5 $O/b.err; exit 1; }
  python -c "
import json
r=json.loads(open('$O/b.json').read().strip().splitlines()[-1])
print('$label', round(r['ms_per_step'],3),'ms')"
}
for rep in 1 2; do
  run "headline default" "A=1" || exit 1
  run "headline unfused-gm" "DVAE_FUSED_GM_MIN=-1" || exit 1
  run "headline fused-gm-4096" "DVAE_FUSED_GM_MIN=4096" || exit 1
  run "c3 default" "A=1" --config c3 || exit 1
  run "c3 unfused-gm" "DVAE_FUSED_GM_MIN=-1" --config c3 || exit 1
  run "c2 default" "A=1" --config c2 || exit 1
  run "c2 unfused-gm" "DVAE_FUSED_GM_MIN=-1" --config c2 || exit 1
  run "B512 small" "A=1" --global-batch 512 --steps 50 || exit 1
  run "B512 nosmall" "DVAE_SMALL_ROWS=0" --global-batch 512 --steps 50 || exit 1
  run "B2048 default" "A=1" --global-batch 2048 --steps 30 || exit 1
  run "B2048 small2048" "DVAE_SMALL_ROWS=2048" --global-batch 2048 --steps 30 || exit 1
done
