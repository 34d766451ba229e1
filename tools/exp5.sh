#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp5
mkdir -p $O
cd $R
# 272 skeleton without parent rows, 784 with (2-byte stores), 262416 = 16|256|262144 parent + grand-parent rows as wide stores, 262480 the same without the grand-parent rows
timeout 300 bash tools/ablate_sweep.sh 272 784 262416 262480 272 784 262416 262480 > $O/ablate.log 2>&1
cat $O/ablate.log | grep -v amdgpu.ids
