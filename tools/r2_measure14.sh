#!/bin/bash
# x6 default: counter passes (headline, c3), then every configuration + the strong-scaling shards
bash tools/pmc2.sh r02x "" --steps 2 --warmup 1 || exit 1
python tools/pmc_summ.py r02x 3 > gpurun_out/pmc_r02x_summary.txt || exit 1
head -14 gpurun_out/pmc_r02x_summary.txt
bash tools/pmc2.sh r02c3 "" --config c3 --steps 1 --warmup 1 || exit 1
python tools/pmc_summ.py r02c3 2 > gpurun_out/pmc_r02c3_summary.txt || exit 1
head -8 gpurun_out/pmc_r02c3_summary.txt
rm -rf gpurun_out/pmc_r02x_* gpurun_out/pmc_r02c3_mfma gpurun_out/pmc_r02c3_fetch gpurun_out/pmc_r02c3_write
bash tools/r2_configs.sh
