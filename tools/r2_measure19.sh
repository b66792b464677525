#!/bin/bash
# final checks of the round: whole GPU suite (x6 core, fp32 core, NaN-poisoned workspaces), then every configuration
O=gpurun_out/r2_22; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all.log 2>&1 || { tail -40 $O/all.log; exit 1; }
tail -1 $O/all.log
DVAE_GEMM_X6=0 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all_f32.log 2>&1 || { tail -40 $O/all_f32.log; exit 1; }
tail -1 $O/all_f32.log
DVAE_POISON_WS=1 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all_poison.log 2>&1 || { tail -40 $O/all_poison.log; exit 1; }
tail -1 $O/all_poison.log
bash tools/r2_configs.sh
timeout -k 10 200 python tools/bench_svae.py > $O/svae.txt 2>&1; tail -5 $O/svae.txt
