#!/bin/bash
# what fused_tail's time is made of: rocprofv3 durations (product and profiling build with parts compiled out) and SQ counters
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-tailprof}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="--no-cpu-baseline --no-end-to-end --no-extras --steps 40 --warmup 10 --spinup-ms 50"
stats() { # name, env..., then the durations per kernel
  local name=$1; shift
  env "$@" rocprofv3 --kernel-trace --stats --output-format csv -d $O/$name -o b -- python $R/tools/bench_dbg.py $B > /dev/null 2> $O/$name.log
  python - $O/$name $name <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        k = "fused_main" if "fused_main" in n else "fused_tail" if "fused_tail" in n else None
        if k: print(sys.argv[2], k, "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 2), "min_us", round(float(r["MinNs"]) / 1e3, 2))
PY
}
stats full BT_FUSED_ABLATE=0
stats skeleton BT_FUSED_ABLATE=1442840576
stats no_lod1_stores BT_FUSED_ABLATE=1073741824
stats no_loads BT_FUSED_ABLATE=33554432
stats lod1_only BT_FUSED_ABLATE=67108864
stats no_aprons BT_FUSED_ABLATE=268435456
# the tail launched twice in a row (profiling build): first launch vs the immediate second one (warm instruction cache, warm L2 / TLB)
stats twice BT_FUSED_ABLATE=0 BT_FUSED_TAIL_TWICE=1
python - $O/twice <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f)))
    tails = [(s, e) for s, e, n in rows if "fused_tail" in n]
    first, second = tails[0::2], tails[1::2]
    avg = lambda v: sum(e - s for s, e in v) / max(len(v), 1) / 1e3
    gaps = [b[0] - a[1] for a, b in zip(first, second)]
    print("twice: first launch avg_us", round(avg(first), 2), "second launch avg_us", round(avg(second), 2), "gap between them avg_us", round(sum(gaps) / max(len(gaps), 1) / 1e3, 2))
    mains = [(s, e) for s, e, n in rows if "fused_main" in n]
    g2 = [t[0] - m[1] for m, t in zip(mains, first)]
    print("gap fused_main end -> fused_tail start avg_us", round(sum(g2) / max(len(g2), 1) / 1e3, 2))
PY
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_WR \
    --kernel-trace --output-format csv -d $O/pmc_sq -o b -- python $R/bench.py $B --steps 3 --warmup 1 > /dev/null 2> $O/pmc_sq.log
python - $O <<'PY'
import collections, csv, glob, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/pmc_sq/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        k = "fused_main" if "fused_main" in n else "fused_tail" if "fused_tail" in n else None
        if k: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v)) for n, v in sorted(c.items())})
PY
