#!/bin/bash
# usage: tools/pmc_svae.sh -- three separate counter passes over a 3-step S-VAE train loop (GPU box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in "mfma:SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" "fetch:FETCH_SIZE" "write:WRITE_SIZE"; do
  name=${pass%%:*}; ctr=${pass#*:}
  rm -rf gpurun_out/pmc_svae_$name
  timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_svae_$name -- python3 tools/prof_svae.py 3 > gpurun_out/pmc_svae_$name.log 2>&1 || { echo "pass $name failed"; tail -3 gpurun_out/pmc_svae_$name.log; exit 1; }
  echo "pass $name ok"
done
