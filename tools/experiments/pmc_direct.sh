#!/bin/bash
# SQ counters of the non-headline configs' kernels (tools/config_bench.py) -> gpurun_out/direct/
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/direct
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD" \
           "SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_SCA" \
           "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_IFETCH SQ_THREAD_CYCLES_VALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT SQ_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_TRANS SQ_IFETCH_LEVEL SQ_ITEMS SQ_ACCUM_PREV"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o b -- python $R/tools/config_bench.py --config2 > /dev/null 2> $O/p$i.log
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "direct" if "fused_direct" in k else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in acc:
    for n, v in sorted(acc[k].items()):
        print(k, n, round(sum(v) / len(v)), len(v))
PY
