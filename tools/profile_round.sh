#!/bin/bash
# Collect the judged profile artefacts of one round on the GPU box:
#   tools/profile_round.sh r06       (run through gpurun from the repo root)
# writes gpurun_out/<tag>/...; tools/pmc_summary.py then condenses them into profiles/<tag>_*.
# Counters are collected in their own passes (one --pmc group per run, kernel trace only), for FIVE workloads: the headline job and
# the parity configurations whose roofline fractions DESIGN.md quotes (masked 16k re-run and fresh, config 2 height / albedo, config 5 cube).
set -u
TAG=${1:-r06}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
# the profiled command runs every job on ONE stream (--pipeline 1): a kernel that shares the GPU with another job's has no
# duration of its own; the roofline of the default (pipelined) bench line is computed from its one-stream pass too
B="python $R/bench.py --no-cpu-baseline --no-end-to-end --no-extras --no-workloads"  # (only the headline job's kernels in the profiled process)
C="python $R/tools/config_bench.py"

python $R/bench.py --verify > $O/bench_n1_verified.json 2> $O/bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- $B > $O/bench_under_rocprofv3.json 2> $O/stats.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS \
    --kernel-trace --output-format csv -d $O/pmc_sq -o bench -- $B --steps 3 --warmup 1 --spinup-ms 0 > /dev/null 2> $O/pmc_sq.log
# memory-side counters, per workload: FETCH_SIZE and WRITE_SIZE (the guide's HBM section: separate passes; FETCH_SIZE x 2 where the
# read requests are 128-byte ones), and the request-size histograms that say whether they are (TCC has four counter slots per pass)
pmc() {  # workload name, then the command
  local w=$1; shift
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc/$w/fetch -o p -- "$@" > /dev/null 2> $O/pmc_${w}_fetch.log
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc/$w/write -o p -- "$@" > /dev/null 2> $O/pmc_${w}_write.log
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --kernel-trace --output-format csv -d $O/pmc/$w/rdreq -o p -- "$@" > /dev/null 2> $O/pmc_${w}_rdreq.log
  rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv -d $O/pmc/$w/wrreq -o p -- "$@" > /dev/null 2> $O/pmc_${w}_wrreq.log
}
pmc headline_16k $B --steps 3 --warmup 1 --spinup-ms 0
for w in config3_masked_16k config3_masked_16k_fresh config2_height_4k config2_albedo_4k config5_cube_height_8k; do
  pmc $w $C --only $w --steps 3
done
# the non-headline BASELINE configs (parity cases) with durations, and the tiling prepass, for the record
rocprofv3 --kernel-trace --stats --output-format csv -d $O/config_stats -o cfg -- $C > $O/config_bench.json 2> $O/config_bench.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/masked_stats -o cfg -- $C --masked16k --fresh > $O/masked16k.json 2> $O/masked16k.log
# the reference's two examples end to end (sources in host memory -> files written), streamed with the serial legs beside them
$C --end-to-end > $O/end_to_end_examples.json 2> $O/end_to_end_examples.log
# the closure record of fused_main: kernel, memory skeleton and linear copy of the byte mix from ONE process (the bench line's roofline block)
python - "$O" <<'PY'
import json, sys
line = json.load(open(sys.argv[1] + "/bench_n1_verified.json"))
json.dump({"ms_per_step": line["ms_per_step"], "roofline": line["roofline"]}, open(sys.argv[1] + "/closure_last_run.json", "w"), indent=1)
PY
python $R/tools/refine_bench.py --sweep > $O/refine_bench.json 2> $O/refine_bench.log
find $O -name '*.csv' | sort | head -60
