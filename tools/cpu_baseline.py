#!/usr/bin/env python3
"""The CPU baseline leg of bench.py, in a process of its own (so that the OpenMP runtime starts with the thread binding
set below): oracle/bt_oracle.c — a PORT, the reference has no CPU path — built on this machine with the flags BASELINE.md §3
states (-O3 -march=native -ffp-contract=off -fopenmp), timed over the span the reference times (preprocessor.rs:363,419:
sources in memory -> all tiles produced, and -> all files written).

  tools/cpu_baseline.py <raster.npy> <out_dir_parent>      prints one JSON object

TEST / BENCH INFRASTRUCTURE ONLY: this is the only kind of place (besides tests/ and smoke()) that may touch oracle/."""
import ctypes as C
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TEXTURE_SIZE, BORDER, ATLAS_SIZE = 512, 2, 2048


def main():
    raster_path, out_parent = sys.argv[1], sys.argv[2]
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O

    hardware = os.cpu_count() or 1
    cores = O.usable_cores()  # the cgroup quota, not the 256 hardware threads the container can see
    if "OMP_PROC_BIND" not in os.environ:  # must be in the environment before libgomp initialises: re-exec once
        env = dict(os.environ, OMP_PROC_BIND="false" if cores < hardware else "spread", OMP_PLACES="threads", OMP_NUM_THREADS=str(cores), OMP_DYNAMIC="false")
        os.execve(sys.executable, [sys.executable] + sys.argv, env)
    import numpy as np

    build = "native (-O3 -march=native -ffp-contract=off -fopenmp, built on this machine)"
    try:
        subprocess.check_call(["make", "-C", O.ORACLE_DIR, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        O._LIB_PATH = os.path.join(O.ORACLE_DIR, "libbt_oracle_native.so")
        O.build = lambda force=False: O._LIB_PATH
    except Exception as e:  # no compiler on this box: the shipped -O2 build
        build = f"the shipped -O2 build (native build failed: {e!r})"
    L = O.lib()
    dp, up = C.POINTER(C.c_double), C.POINTER(C.c_uint32)
    L.orc_run_blocks.argtypes = [C.c_void_p, C.c_int, C.c_uint32, dp, up, C.c_uint32, up]
    L.orc_atlas_touch.argtypes = [C.c_void_p, C.c_int]

    src = np.load(raster_path, mmap_mode="r")
    size = src.shape[0]
    # bounded sample: the whole workload where >= 8 cores make it a matter of seconds, else the top-left 8192^2 window
    sample, lods = (size, 6) if cores >= 8 else (min(size, 8192), 5)
    window = np.ascontiguousarray(src[:sample, :sample])

    def run(win, nlods, threads, rows_per_block):
        a = O.OracleAtlas(nlods, ATLAS_SIZE, False, [(TEXTURE_SIZE, BORDER, 1, O.FORMAT_R16)])
        a.preprocess_tile(0, win, (0, nlods))
        L.orc_atlas_touch(a._h, threads)
        secs, tasks, n = (C.c_double * 64)(), (C.c_uint32 * 64)(), C.c_uint32()
        t0 = time.perf_counter()
        rc = L.orc_run_blocks(a._h, threads, rows_per_block, secs, tasks, 64, C.byref(n))
        dt = time.perf_counter() - t0
        assert rc == 0
        return a, dt, [(int(tasks[i]), float(secs[i])) for i in range(min(n.value, 64))]

    run(np.ascontiguousarray(window[:2048, :2048]), 3, cores, 32)  # thread pool up, code paged in
    a, dt, phases = run(window, lods, cores, 32)
    tiles = len(a.tiles())
    out_dir = tempfile.mkdtemp(prefix="bt_cpu_baseline_", dir=out_parent)
    t0 = time.perf_counter()
    a.save_attachment(0, out_dir)
    a.save_tile_config(os.path.join(out_dir, "config.tc"))
    dt_files = time.perf_counter() - t0
    shutil.rmtree(out_dir, ignore_errors=True)
    # one thread, the same schedule, on a window a single core finishes in seconds
    small = np.ascontiguousarray(window[:4096, :4096])
    a1, dt1, _ = run(small, 4, 1, 32)
    tiles1 = len(a1.tiles())
    what = "the whole workload" if sample == size else f"the top-left {sample}x{sample} window of the same heightmap"
    one = tiles1 / dt1
    print(json.dumps({
        "value": tiles / dt, "unit": "tiles/s", "cores": cores, "kind": "port",
        "cores_note": f"{cores} = CPUs this process may use (affinity mask capped by the cgroup CPU quota); the box shows {hardware} hardware threads",
        "sample": f"oracle/bt_oracle.c, {build}; units = (task, 32-row block) over OpenMP, threads bound when the process owns the machine (OMP_PROC_BIND), {cores} threads, atlas "
                  f"pages touched before the clock starts; {what}, lod_count {lods}: {tiles} tiles of 512^2 in {dt:.2f} s",
        "speedup_over_one_thread": (tiles / dt) / one, "parallel_efficiency": (tiles / dt) / one / cores,
        "phases": [{"tasks": t, "seconds": round(s, 4)} for t, s in phases],
        "phases_note": "queue order: split (finest tiles), downsample per LOD, stitch per LOD; the phases of 16, 4 and 1 tasks are the "
                       "top of the pyramid — their units are the row blocks of those few tiles",
        "files_written": {"value": tiles / (dt + dt_files), "unit": "tiles/s", "cores": cores,
                          "sample": f"the same run + its {tiles} .bin files and config.tc written under {out_parent} ({dt_files:.2f} s, one thread, "
                                    f"like the reference's save tasks: one fs::write per tile)"},
        "one_thread": {"value": one, "unit": "tiles/s", "cores": 1,
                       "sample": f"the top-left 4096x4096 window, lod_count 4, the same build and schedule: {tiles1} tiles in {dt1:.2f} s"}}))


if __name__ == "__main__":
    main()
