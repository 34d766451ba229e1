#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into profiles/<tag>_*.

  python tools/pmc_summary.py r01          (reads gpurun_out/r01/, writes profiles/r01_*)

HBM traffic per launch is corrected as MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts exactly half the bytes of a wide coalesced
read stream -> x2; WRITE_SIZE as reported.  Warm-up launches are included in the averages: every launch of
a kernel does identical work in this bench.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ("fused_main", "fused_tail", "fused_corner", "fused_todo", "split", "downsample", "stitch")


def kind_of(kernel_name):
    for k in KINDS:
        if f"{k}_kernel" in kernel_name:
            return k
    return None


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    summary = collections.defaultdict(dict)
    for group in ("fetch", "write", "sq"):
        files = glob.glob(os.path.join(src, f"pmc_{group}", "**", "*counter_collection.csv"), recursive=True)
        if not files:
            continue
        shutil.copy(files[0], os.path.join(dst, f"{tag}_pmc_{group}_counter_collection.csv"))
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for row in csv.DictReader(open(files[0])):
            k = kind_of(row["Kernel_Name"])
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, counters in acc.items():
            for name, values in counters.items():
                summary[k][name] = {"avg_per_launch": sum(values) / len(values), "launches": len(values)}
    for k, s in summary.items():
        if "FETCH_SIZE" in s and "WRITE_SIZE" in s:
            wr = s["WRITE_SIZE"]["avg_per_launch"] * 1024
            if k == "fused_main":  # 16-byte-per-lane streaming loads: the access width the guide's x2 correction is calibrated for
                rd = s["FETCH_SIZE"]["avg_per_launch"] * 1024 * 2
                s["hbm_traffic_bytes"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr}
            else:  # narrower loads: FETCH_SIZE is uncalibrated on gfx950 — raw counter only, no traffic claim
                s["hbm_counters_raw_bytes"] = {"FETCH_SIZE_raw": s["FETCH_SIZE"]["avg_per_launch"] * 1024, "WRITE_SIZE": wr,
                                               "note": "uncalibrated for this kernel's access widths; not a traffic figure"}
    out = dict(summary)
    out["_note"] = ("HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3 reports "
                    "FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE counts exactly half the bytes of a wide coalesced "
                    "read stream -> x2; WRITE_SIZE as reported (uncalibrated)")
    import subprocess

    try:
        out["_git"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        out["_git"] = None
    out["_command"] = "tools/profile_round.sh (one rocprofv3 --pmc pass per counter group, --kernel-trace only)"
    out["_workload"] = "synthetic 16384x16384 fBm R16, T=512, b=2, lod_count=6 (1365 tiles), fused path, 1x MI355X"
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)

    stats = glob.glob(os.path.join(src, "stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(dst, f"{tag}_bench_kernel_stats.csv"))
    cfg = glob.glob(os.path.join(src, "config_stats", "**", "*kernel_stats.csv"), recursive=True)
    if cfg:
        shutil.copy(cfg[0], os.path.join(dst, f"{tag}_config_bench_kernel_stats.csv"))
    for name in ("bench_n1_verified.json", "bench_under_rocprofv3.json", "config_bench.json", "refine_bench.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
    for k in sorted(summary):
        t = summary[k].get("hbm_traffic_bytes")
        print(k, {n: round(v["avg_per_launch"]) for n, v in summary[k].items() if isinstance(v, dict) and "avg_per_launch" in v}, t)


if __name__ == "__main__":
    main()
