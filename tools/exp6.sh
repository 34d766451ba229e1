#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp6
mkdir -p $O
cd $R
# finest rows as one 16-byte-per-lane store per wave and quad: alone (shifted / aligned), with the loads, with loads + wide parent rows; 272 = reference (loads + dword finest stores)
timeout 300 bash tools/ablate_sweep.sh 1048856 1052952 1048848 1052944 1310992 272 1048856 1052952 1048848 1052944 1310992 272 > $O/ablate.log 2>&1
cat $O/ablate.log | grep -v amdgpu.ids
