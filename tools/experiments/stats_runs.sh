#!/bin/bash
# the rocprofv3 --stats pass of tools/profile_round.sh N times (one process each): fused_main lands in one of two modes per process
# (profiles/r03_fused_main_experiments.txt §11), so one run is not the kernel.  tools/experiments/stats_runs.sh <tag> [N]
set -u
TAG=${1:-r03}; N=${2:-3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
for i in $(seq 1 $N); do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_run$i -o bench -- python $R/bench.py --no-cpu-baseline --no-end-to-end --no-extras > $O/bench_under_rocprofv3_run$i.json 2> $O/stats_run$i.log
  grep fused_main $O/stats_run$i/bench_kernel_stats.csv | cut -d, -f1-4 | cut -c1-160
done
