#!/bin/bash
# round-2 experiment batch (profiling build): workgroup -> tile order search + parent-store ablations of fused_main
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp3
mkdir -p $O
cd $R
timeout 600 python tools/order_search.py --out $O/order_search.json > $O/order_search.log 2>&1
tail -20 $O/order_search.log
# 0 complete, 4 no parent stores, 64 no grand-parent stores, 68 neither, 784 memory skeleton, 66320 skeleton with parent bursts, 16 prologue + loop only
timeout 300 bash tools/ablate_sweep.sh 0 4 64 68 784 66320 16 0 > $O/ablate.log 2>&1
cat $O/ablate.log
