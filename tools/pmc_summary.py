#!/usr/bin/env python3
"""Condense the rocprofv3 output of tools/profile_round.sh into profiles/<tag>_*.

  python tools/pmc_summary.py r05          (reads gpurun_out/r05/, writes profiles/r05_*)

HBM traffic per launch, per (workload, kernel), corrected as MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3 reports
FETCH_SIZE / WRITE_SIZE in KiB; on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B, i.e. exactly half the bytes of a read stream
whose requests are 128-byte ones -> x 2; WRITE_SIZE as reported.  Whether a kernel's requests ARE 128-byte ones is read from the
request-size histogram collected in the same round (TCC_EA0_RDREQ_{32B,64B,128B}): `read_from_request_sizes` is the byte count
that histogram gives on its own (32 n32 + 64 n64 + 128 n128) and must agree with the corrected FETCH_SIZE; likewise
64 n64 + 32 (n - n64) for the writes.  Only summaries are kept (the raw per-dispatch CSVs stay in gpurun_out/).
Warm-up launches are included in the averages: every launch of a kernel does identical work in these benches.
"""
import collections
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KINDS = ("fused_main", "fused_tail2", "fused_tail", "fused_direct_rgba8", "fused_corner", "stitch_region", "split", "downsample", "stitch")
ALIAS = {"fused_tail2": "fused_tail", "fused_direct_rgba8": "fused_direct", "stitch_region": "stitch"}
WORKLOADS = {"headline_16k": "synthetic 16384^2 fBm R16 (seed 42), lod_count 6, 1365 tiles (bench.py)",
             "config3_masked_16k": "the same with the 5 % no-data mask (seed 43), re-runs of a kept queue (previous values are fetched)",
             "config3_masked_16k_fresh": "the same, every run on an atlas nothing has written since bt_atlas_create (prev_zero: no previous-value fetches)",
             "config2_height_4k": "4096^2 R16, lod_count 4, 85 tiles",
             "config2_albedo_4k": "4096^2 Rgba8, lod_count 4, 85 tiles",
             "config5_cube_height_8k": "6 x 8192^2 R16 faces, lod_count 5, 2046 tiles"}


def kind_of(kernel_name):
    for k in KINDS:
        if f"{k}_kernel" in kernel_name:
            return ALIAS.get(k, k)
    return None


def counters_of(directory):
    """{kernel kind: {counter: average per launch}} of every counter_collection.csv under `directory`"""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(directory, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            k = kind_of(row["Kernel_Name"])
            if k:
                acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    return {k: {n: {"avg_per_launch": sum(v) / len(v), "launches": len(v)} for n, v in c.items()} for k, c in acc.items()}


def traffic(c):
    """the HBM byte counts of one (workload, kernel) from its counters"""
    out = {}
    avg = lambda n: c[n]["avg_per_launch"] if n in c else None  # noqa: E731
    if avg("FETCH_SIZE") is not None and avg("WRITE_SIZE") is not None:
        rd, wr = avg("FETCH_SIZE") * 1024 * 2, avg("WRITE_SIZE") * 1024
        out["hbm_traffic_bytes"] = {"read_corrected_x2": rd, "write": wr, "total": rd + wr}
    if avg("TCC_EA0_RDREQ_sum") is not None:
        n, n32, n64, n128 = avg("TCC_EA0_RDREQ_sum"), avg("TCC_EA0_RDREQ_32B_sum") or 0.0, avg("TCC_EA0_RDREQ_64B_sum") or 0.0, avg("TCC_EA0_RDREQ_128B_sum") or 0.0
        out["read_requests"] = {"all": n, "32B": n32, "64B": n64, "128B": n128, "share_128B": n128 / n if n else None,
                                "read_from_request_sizes": 32 * n32 + 64 * n64 + 128 * n128}
    if avg("TCC_EA0_WRREQ_sum") is not None:
        n, n64 = avg("TCC_EA0_WRREQ_sum"), avg("TCC_EA0_WRREQ_64B_sum") or 0.0
        out["write_requests"] = {"all": n, "64B": n64, "share_64B": n64 / n if n else None, "write_from_request_sizes": 64 * n64 + 32 * (n - n64)}
    return out


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    src = os.path.join(ROOT, "gpurun_out", tag)
    dst = os.path.join(ROOT, "profiles")
    # algorithmic bytes per (workload, kernel) from the benches' own launch profiles
    algorithmic = collections.defaultdict(dict)
    try:
        line = json.load(open(os.path.join(src, "bench_under_rocprofv3.json")))
        for l in line["config"]["launches"]:
            algorithmic["headline_16k"][l["kind"]] = l["algorithmic_bytes"]
    except (OSError, ValueError, KeyError):
        pass
    for name in ("config_bench.json", "masked16k.json", "end_to_end_examples.json"):
        try:
            for w, rec in json.load(open(os.path.join(src, name))).items():
                for l in rec.get("launches") or []:
                    algorithmic[w][l[0]] = l[2]
        except (OSError, ValueError, KeyError, IndexError):
            pass

    workloads = {}
    for w in WORKLOADS:
        per_kernel = counters_of(os.path.join(src, "pmc", w))
        if not per_kernel:
            continue
        workloads[w] = {"_workload": WORKLOADS[w]}
        for k, c in per_kernel.items():
            rec = dict(c)
            rec.update(traffic(c))
            alg = algorithmic.get(w, {}).get(k)
            if alg and "hbm_traffic_bytes" in rec:
                rec["algorithmic_bytes"] = alg
                rec["traffic_over_algorithmic"] = rec["hbm_traffic_bytes"]["total"] / alg
            workloads[w][k] = rec
    sq = counters_of(os.path.join(src, "pmc_sq"))

    # top level = the headline workload's kernels (what bench.py's `roofline.traffic` reads), + the SQ counters of the same command
    out = {}
    for k, rec in workloads.get("headline_16k", {}).items():
        if not k.startswith("_"):
            out[k] = dict(rec)
    for k, c in sq.items():
        out.setdefault(k, {}).update(c)
    out["workloads"] = workloads
    out["_note"] = ("HBM traffic per launch, corrected as MI355X_MICROARCH.md (HBM section) prescribes: rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; "
                    "on gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B = half the bytes of a stream of 128-byte read requests -> x 2 (the request-size "
                    "histograms under `read_requests` say how far that holds per kernel); WRITE_SIZE as reported")
    try:
        out["_git"] = subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], text=True).strip()
    except Exception:
        out["_git"] = None
    out["_command"] = "tools/profile_round.sh (one rocprofv3 --pmc pass per counter group and workload, --kernel-trace only)"
    out["_workload"] = WORKLOADS["headline_16k"] + ", fused path, 1x MI355X"
    json.dump(out, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1, sort_keys=True)

    for sub, name in (("stats", "bench_kernel_stats.csv"), ("config_stats", "config_bench_kernel_stats.csv"), ("masked_stats", "masked16k_kernel_stats.csv")):
        files = glob.glob(os.path.join(src, sub, "**", "*kernel_stats.csv"), recursive=True)
        if files:
            shutil.copy(files[0], os.path.join(dst, f"{tag}_{name}"))
    for name in ("bench_n1_verified.json", "bench_under_rocprofv3.json", "config_bench.json", "masked16k.json", "refine_bench.json", "end_to_end_examples.json",
                 "closure_last_run.json"):
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, f"{tag}_{name}"))
    for w, rec in workloads.items():
        for k, r in rec.items():
            if k.startswith("_"):
                continue
            t = r.get("hbm_traffic_bytes")
            print(w, k, "traffic", round(t["total"]) if t else None, "alg", r.get("algorithmic_bytes"), "ratio", round(r.get("traffic_over_algorithmic", 0), 3),
                  "128B share", round((r.get("read_requests") or {}).get("share_128B") or 0, 3))


if __name__ == "__main__":
    main()
