#!/bin/bash
# Collect the judged profile artefacts of one round on the GPU box:
#   tools/profile_round.sh r01       (run through gpurun from the repo root)
# writes gpurun_out/<tag>/...; tools/pmc_summary.py then condenses them into profiles/<tag>_*.
# Counters are collected in their own passes (one --pmc group per run, kernel trace only).
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# the profiled command runs every job on ONE stream (--pipeline 1): a kernel that shares the GPU with another job's has no
# duration of its own; the roofline of the default (pipelined) bench line is computed from its one-stream pass too
B="python $R/bench.py --no-cpu-baseline --no-end-to-end --no-extras"

python $R/bench.py --verify > $O/bench_n1_verified.json 2> $O/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_under_rocprofv3.json 2> $O/stats.log
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o bench -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2> $O/pmc_fetch.log
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o bench -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2> $O/pmc_write.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d $O/pmc_sq -o bench -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2> $O/pmc_sq.log
# the non-headline BASELINE configs (parity cases) and the tiling prepass, for the record
rocprofv3 --kernel-trace --stats --output-format csv -d $O/config_stats -o cfg -- python $R/tools/config_bench.py > $O/config_bench.json 2> $O/config_bench.log
python $R/tools/refine_bench.py --sweep > $O/refine_bench.json 2> $O/refine_bench.log
find $O -name '*.csv' | sort
