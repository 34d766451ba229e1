#!/bin/bash
# L2 / fabric-side (TCC) counter passes over the bench (gpurun): which counters exist is read from `rocprofv3 -L`
# on the box, the wish list below is packed four per pass (TCC has 4 slots).  Results: gpurun_out/tcc/summary.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/tcc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters.txt 2>&1
B="python $R/bench.py --no-cpu-baseline --no-end-to-end --steps 3 --warmup 1 --spinup-ms 0"
python - "$O" <<'PY' > $O/groups.txt
import re, sys
txt = open(sys.argv[1] + "/counters.txt").read()
names = set(re.findall(r"Counter_Name\s*:\s*(\S+)", txt))
wish = ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_STALL_sum", "TCC_EA0_WR_UNCACHED_32B_sum",
        "TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_EA0_RDREQ_sum",
        "TCC_EA0_RDREQ_32B_sum", "TCC_WRITE_sum", "TCC_READ_sum", "TCC_WRITEBACK_sum",
        "TCC_NORMAL_WRITEBACK_sum", "TCC_NORMAL_EVICT_sum", "TCC_ALL_TC_OP_WB_WRITEBACK_sum", "TCC_TAG_STALL_sum",
        "TCC_EA0_WRREQ_DRAM_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", "TCC_TOO_MANY_EA_WRREQS_STALL_sum",
        "TCC_EA0_WRREQ_IO_CREDIT_STALL_sum", "TCC_EA0_WRREQ_GMI_CREDIT_STALL_sum", "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", "TCC_SRC_FIFO_FULL_sum",
        "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_ATOMIC_sum", "TCC_BUBBLE_sum",
        "TCC_WRREQ_STALL_max", "TCC_EA0_WRREQ_STALL_max", "TCC_STREAMING_REQ_sum", "TCC_NC_REQ_sum",
        "TCP_TCC_WRITE_REQ_sum", "TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum",
        "TCP_TCC_NC_WRITE_REQ_sum", "TCP_TCC_UC_WRITE_REQ_sum", "TCP_TCC_CC_WRITE_REQ_sum", "TCP_TCC_RW_WRITE_REQ_sum",
        "TCP_TA_TCP_STATE_READ_sum", "TCP_TCR_TCP_STALL_CYCLES_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_GATE_EN1_sum",
        "TCC_EA0_WRREQ_WRITE_DRAM_sum", "TCC_EA0_WRREQ_WRITE_DRAM_32B_sum", "TCC_WRITE_SECTORS_sum", "TCC_READ_SECTORS_sum",
        "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum", "TCC_LATENCY_FIFO_FULL_sum", "TCC_IB_STALL_sum",
        "TCC_BUSY_sum", "TCC_CYCLE_sum", "TCC_IB_REQ_sum", "TCC_BYPASS_REQ_sum",
        "TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCP_TA_ADDR_STALL_CYCLES_sum", "TCP_RFIFO_STALL_CYCLES_sum",
        "TD_STORE_WAVEFRONT_sum", "TD_LOAD_WAVEFRONT_sum", "TD_TC_STALL_sum", "TD_TD_BUSY_sum",
        "TA_BUSY_avr", "TA_BUFFER_WRITE_WAVEFRONTS_sum", "TA_FLAT_WRITE_WAVEFRONTS_sum", "TA_DATA_STALLED_BY_TC_CYCLES_sum"]
have = [w for w in wish if w in names]
missing = [w for w in wish if w not in names]
print("#missing " + " ".join(missing))
groups = {}
for w in have:   # one hardware block per pass, four counters each
    blk = w.split("_")[0]
    groups.setdefault(blk, []).append(w)
for blk, ws in groups.items():
    for i in range(0, len(ws), 4):
        print(" ".join(ws[i:i + 4]))
PY
i=0
grep -v '^#' $O/groups.txt | while read set; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o b -- $B > /dev/null 2> $O/p$i.log || echo "pass $i failed: $set" >> $O/failed.txt
done
python - "$O" <<'PY'
import collections, csv, glob, json, sys
O = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(O + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        k = "fused_main" if "fused_main" in k else "fused_tail" if "fused_tail" in k else "fused_todo" if "fused_todo" in k else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: {n: sum(v) / len(v) for n, v in sorted(c.items())} for k, c in acc.items()}
out["_missing"] = open(O + "/groups.txt").readline().strip()
json.dump(out, open(O + "/summary.json", "w"), indent=1, sort_keys=True)
for n, v in out.get("fused_main", {}).items():
    print("main", n, round(v))
PY
cat $O/failed.txt 2>/dev/null
