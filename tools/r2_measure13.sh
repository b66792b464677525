#!/bin/bash
# x6 default: whole GPU suite (plain + NaN-poisoned workspaces), then the headline profile
O=gpurun_out/r2_17; mkdir -p $O
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all.log 2>&1 || { tail -40 $O/all.log; exit 1; }
tail -1 $O/all.log
DVAE_POISON_WS=1 timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/all_poison.log 2>&1 || { tail -40 $O/all_poison.log; exit 1; }
tail -1 $O/all_poison.log
bash tools/r2_profile.sh
