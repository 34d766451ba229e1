#!/bin/bash
# gpurun helper: the round-3 copy-floor sweep with the lease's clocks recorded next to it.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/copy_floor3
mkdir -p $O
cd $R
{
  echo "== rocm-smi before =="; rocm-smi --showclocks --showperflevel --showpower 2>&1 | grep -v "^$" | head -40
  echo "== sweep =="; timeout 600 tools/copy_floor3.out "$@"
  echo "== rocm-smi after =="; rocm-smi --showclocks --showpower 2>&1 | grep -v "^$" | head -40
} > $O/copy_floor3.txt 2>&1
# clocks while a copy loop runs
( timeout 20 tools/copy_floor3.out quick > /dev/null 2>&1 & sleep 6; echo "== rocm-smi under load =="; rocm-smi --showclocks 2>&1 | grep -v "^$" | head -30; wait ) >> $O/copy_floor3.txt 2>&1
grep -E "TB/s" $O/copy_floor3.txt | sort -k 2 -t: -n | awk '{print}' | sort -t: -k2 | tail -5
grep -E "copy" $O/copy_floor3.txt | awk '{print $NF, $0}' | sort -k2,2 -n -r | sort -k1,1 -n -r -s | head -12
