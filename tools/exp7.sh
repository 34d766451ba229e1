#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/exp7
mkdir -p $O
cd $R
# 0: parent rows through LDS (wide stores); 2097152: straight from registers (the former form)
timeout 300 bash tools/ablate_sweep.sh 0 2097152 0 2097152 > $O/ablate.log 2>&1
cat $O/ablate.log | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_gpu_preprocess.py tests/test_golden.py tests/test_gpu_random_sweep.py -m gpu -x -q > $O/tests.log 2>&1
tail -5 $O/tests.log
