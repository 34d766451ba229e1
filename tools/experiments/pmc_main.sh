#!/bin/bash
# Extra counter passes for fused_main (one group per run, kernel trace only): issue activity, LDS, the vector-memory path.
# tools/experiments/pmc_main.sh  (through gpurun, from the repo root) -> gpurun_out/pmc_main/<group>/
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_main
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="python $R/bench.py --no-cpu-baseline --no-end-to-end --no-extras --steps 3 --warmup 1 --spinup-ms 0"
run() { timeout 150 rocprofv3 --pmc "${@:2}" --kernel-trace --output-format csv -d $O/$1 -o bench -- $B > /dev/null 2> $O/$1.log; }
run issue SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CU_CYCLES
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
run vmem SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
# (the TA_* and TCP_* groups abort rocprofv3 on this image — signal 6 — and then sit until killed: 15 GPU-minutes lost once; not collected)
# run ta TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS TA_FLAT_READ_LDS_WAVEFRONTS TA_FLAT_READ_WAVEFRONTS TA_BUFFER_TOTAL_CYCLES TA_TOTAL_WAVEFRONTS
# run tcp TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ_LATENCY TCP_TCC_READ_REQ_LATENCY TCP_TCC_WRITE_REQ TCP_TCC_READ_REQ TCP_TCR_TCP_STALL_CYCLES TCP_TA_TCP_STATE_READ TCP_GATE_EN1
find $O -name '*counter_collection.csv' | sort
