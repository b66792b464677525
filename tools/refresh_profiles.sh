#!/bin/bash
# usage: tools/refresh_profiles.sh <tag> "<build note>" -- copy the end-of-round set of `tools/measure.sh final <tag>` (merged into
# gpurun_out/<tag>/) into profiles/r03_* with a header that names the build; run in the build container after the GPU call
tag=$1; note=$2
cd "$(dirname "$0")/.." || exit 1
O=gpurun_out/$tag
cp $O/bench.json profiles/r03_bench.json
cp $O/configs.jsonl profiles/r03_configs.jsonl
cp $O/shards.jsonl profiles/r03_shards.jsonl
{ echo "# headline step, $note; rocprofv3 --kernel-trace, tools/kernel_summary.py + tools/conc.py"; cat $O/kernel_summary.txt; } > profiles/r03_kernel_trace_summary.txt
{ echo "# headline step, $note: three separate rocprofv3 --pmc passes (tools/pmc2.sh), per train step; mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (4 SQ_BUSY_CU_CYCLES); fetch = 2 x FETCH_SIZE (gfx950 note), write = WRITE_SIZE"; cat $O/pmc_summary.txt; } > profiles/r03_pmc_summary.txt
python tools/pmc_traffic.py headline $tag 3
