#!/bin/bash
# usage: tools/pmc2.sh <tag> "<ENV=.. assignments or empty>" <bench args...>
# three separate counter passes (MFMA/LDS | FETCH_SIZE | WRITE_SIZE) over a short bench run (GPU box)
tag=$1; envs=$2; shift 2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for e in $envs; do export "$e"; done
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rm -rf gpurun_out/pmc_${tag}_$name
  timeout -s KILL 300 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_${tag}_$name -- python3 bench.py --no-cpu-baseline --no-profile --no-core-ab "$@" > gpurun_out/pmc_${tag}_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_${tag}_$name.log; exit 1; }
  echo "$tag pass $name ok"
done
