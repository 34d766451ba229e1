R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/tlb; rm -rf $O; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for cfg in lin3w k0 k2; do
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum --kernel-trace --output-format csv -d $O/$cfg -o b -- $R/tools/copy_floor2.out $cfg > /dev/null 2> $O/$cfg.log
  rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum --kernel-trace --output-format csv -d $O/${cfg}_b -o b -- $R/tools/copy_floor2.out $cfg > /dev/null 2> $O/${cfg}_b.log
done
python - <<PY
import csv, glob, collections
for cfg in ("lin3w", "k0", "k2"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$O/%s*/**/*counter_collection.csv" % cfg, recursive=True):
        for r in csv.DictReader(open(f)):
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(cfg, {k: round(sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
