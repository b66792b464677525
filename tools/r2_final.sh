#!/bin/bash
# end-of-round artefacts: headline bench line + kernel trace, small-shard traces, counter passes
bash tools/r2_profile.sh || exit 1
bash tools/r2_measure10.sh > gpurun_out/r2_small_traces.txt 2>&1 || { tail -5 gpurun_out/r2_small_traces.txt; exit 1; }
grep -h "^# " gpurun_out/r2_14/trace_b512.txt gpurun_out/r2_14/trace_b32.txt
bash tools/pmc2.sh r02x "" --steps 2 --warmup 1 && python tools/pmc_summ.py r02x 3 > gpurun_out/r02x_pmc_summary.txt && rm -rf gpurun_out/pmc_r02x_mfma gpurun_out/pmc_r02x_fetch gpurun_out/pmc_r02x_write && head -8 gpurun_out/r02x_pmc_summary.txt
