#!/usr/bin/env python3
"""Headline benchmark: terrain tiles/s preprocessed on a synthetic 16384^2 heightmap (BASELINE.json).

One step = one full pass of the hot path (split -> LOD pyramid -> border stitch -> atlas packing)
over the resident 16k x 16k R16 heightmap into 1365 tiles of 512^2 (T=512, b=2, lod_count=6).
Inputs are already in HBM when the timed region starts; value = tiles / s over all ranks.

  python bench.py [--gpus N] [--steps K] [--warmup W]
The headline (`value`, `ms_per_step`) is defined the same way at every N: ONE job at a time on ONE stream per rank.
N = 1: job after job on the context's stream.  A second pass with two independent jobs in flight (two contexts = two HIP
streams, an atlas each, the source shared, steps issued round-robin) is reported as `config.ms_per_step_two_in_flight`.
N > 1: one rank per GPU (launched by torch.distributed.run, or — when WORLD_SIZE is not set — by this script re-executing
itself under it): the finest tile grid is split into column strips, every rank preprocesses its strip, ONE grouped RCCL
collective per step issued by the library assembles the atlas on EVERY rank (in-place all-gathers, `result: replicated`,
BASELINE config 4), then the short finishing kernels run on every rank (strong scaling: the 16k job is fixed).  The same run
also reports, as extras in `config`: the kernels alone, the collective alone, and the `distributed` result (the finest LOD
stays on the rank that computed it; a quarter of the bytes travel).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

SIZE = 16384
TEXTURE_SIZE, BORDER, LOD_COUNT, ATLAS_SIZE = 512, 2, 6, 2048
SEED = 42
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s HBM3E
HBM_ACHIEVABLE_GBS = 6290.0  # the same guide's measured float4 copy (line 35); never the `peak` of the roofline block


def ram_directory():
    """a RAM-backed directory when it has room (the container's overlay disk throttles at its dirty-page limit), else None"""
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > (2 << 30):
            return "/dev/shm"
    except OSError:
        pass
    return None


def cpu_baseline(device, src_ptr):
    """The CPU oracle (a port: the reference has no CPU path and cannot be built here) timed on the host cores over the
    span the reference times (preprocessor.rs:363,419): tools/cpu_baseline.py in a process of its own — built on this machine
    with BASELINE.md §3's flags, threads bound, atlas pages pre-touched, per-phase seconds, files on the same file system as the
    product's end_to_end leg.  A bounded sample: the whole 16k workload with >= 32 cores (a few seconds), else a window."""
    import subprocess
    import tempfile

    import numpy as np

    host = device.download(src_ptr, (SIZE, SIZE), np.uint16)
    parent = ram_directory() or tempfile.gettempdir()
    path = os.path.join(parent, f"bt_bench_source_{os.getpid()}.npy")
    np.save(path, host)
    del host
    try:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "cpu_baseline.py"), path, parent], capture_output=True, text=True, timeout=900)
        if out.returncode != 0:
            return {"error": out.stderr[-1500:]}
        return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    finally:
        try:
            os.remove(path)
        except OSError:
            pass


def oracle_atlas(device, src_ptr):
    """--verify: the 16k job through the checker — the oracle's queue driver with the reference's own WGSL executed for
    every task (oracle/_ref) when that library is present, else the oracle's own kernels"""
    import numpy as np

    import _oracle as O

    host = device.download(src_ptr, (SIZE, SIZE), np.uint16)
    a = O.OracleAtlas(LOD_COUNT, ATLAS_SIZE, False, [(TEXTURE_SIZE, BORDER, 1, O.FORMAT_R16)])
    checker = "oracle/bt_oracle.c"
    try:
        import _wgslref as W

        if W.available():
            W.attach(a)
            checker = "oracle/_ref: the reference's WGSL executed on the CPU (queue driven by oracle/bt_oracle.c)"
    except Exception:
        pass
    a.preprocess_tile(0, host, (0, LOD_COUNT))
    a.run(O.usable_cores())
    return a, (SIZE, LOD_COUNT), checker


def end_to_end(device, src_ptr):
    """The reference's own timing span (preprocessor.rs:363,419: all sources loaded -> all saves done) for the
    product: source raster in ordinary host memory -> H2D -> kernels -> D2H -> every .bin tile file + config.tc
    written.  Not part of `value` (that is device-resident); reported so that the PCIe / file-system side is measured
    rather than estimated."""
    import shutil
    import tempfile

    import numpy as np

    import bevy_terrain_amd as bt

    host = device.download(src_ptr, (SIZE, SIZE), np.uint16)  # the "loaded source image" the span starts from
    cfg = bt.TerrainConfig(lod_count=LOD_COUNT, atlas_size=ATLAS_SIZE, path="terrains/bench16k_e2e",
                           model=bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=TEXTURE_SIZE, border_size=BORDER,
                                           format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    device.synchronize()
    # where the files go: a RAM-backed file system when it has room (isolates the library's D2H + write pipeline from
    # the box's disk: the container's overlay disk sustains ~3 GB/s once the kernel's dirty-page limit is reached, and
    # how soon that happens depends on what ran before), else the default temporary directory
    parent = ram_directory()
    root = tempfile.mkdtemp(prefix="bt_e2e_", dir=parent)

    def one_pass(window, lods):
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
        t0 = time.perf_counter()
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="host", lod_range=range(0, lods)),
                            bt.AssetServer().insert("host", window), atlas)
        t1 = time.perf_counter()
        pre.run(atlas)
        t2 = time.perf_counter()
        pre.save(atlas, root)
        t3 = time.perf_counter()
        pre.close()
        return (t1 - t0, t2 - t1, t3 - t2)

    def streamed_pass(window, lods, directory):
        """the overlapped pipeline (bt_preprocessor_run_streamed): deferred upload in bands || kernels || D2H + file writes"""
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, directory)
        t0 = time.perf_counter()
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="host", lod_range=range(0, lods)),
                            bt.AssetServer().insert("host", window), atlas, defer_upload=True)
        st = pre.run_streamed(atlas, directory)
        t1 = time.perf_counter()
        pre.close()
        return t1 - t0, st

    def digest(directory):
        import hashlib

        d = atlas.attachment_directory(directory, 0)
        h = hashlib.sha256()
        for f in sorted(os.listdir(d)):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
        return h.hexdigest(), len(os.listdir(d))

    warm = one_pass(np.ascontiguousarray(host[:2048, :2048]), 3)  # 21 tiles: allocates the pinned staging buffers, warms the paths
    results = [warm, one_pass(host, LOD_COUNT)]
    serial_digest = digest(root)
    streamed = None
    try:
        warm_dir = tempfile.mkdtemp(prefix="bt_e2e_warm_", dir=parent)  # (its own directory: `root` keeps the serial pass's 1365 files for load_back)
        try:
            streamed_pass(np.ascontiguousarray(host[:4096, :4096]), 4, warm_dir)  # warm the side streams / the saver thread's paths
        finally:
            shutil.rmtree(warm_dir, ignore_errors=True)
        # every pass writes a FRESH directory, like a real preprocess run (rewriting the files of the pass before makes the file
        # system truncate 716 MB of pages inside the timed span: passes 3 - 5 of the first round-5 profile took 29 - 30 ms against 19.6 - 20 for
        # the first two); the directory of a pass is removed outside its span
        def fresh_pass():
            d = tempfile.mkdtemp(prefix="bt_e2e_pass_", dir=parent)
            try:
                dt, st = streamed_pass(host, LOD_COUNT, d)
                return dt, st, (digest(d) if fresh_pass.want_digest else None)
            finally:
                shutil.rmtree(d, ignore_errors=True)

        fresh_pass.want_digest = False
        times = []
        for k in range(5):
            fresh_pass.want_digest = k == 4
            dt, st, last_digest = fresh_pass()
            times.append(dt)
        fresh_pass.want_digest = False
        ordered = sorted(times)
        streamed = {"ms": ordered[len(ordered) // 2] * 1e3, "ms_min": ordered[0] * 1e3, "ms_max": ordered[-1] * 1e3, "ms_all": [t * 1e3 for t in times],
                    "passes": len(times), "bands": st["bands"], "overlapped": st["streamed"],
                    "writer_threads": device.io_threads(),  # automatic: min(16, CPUs the process may use) — bt_ctx_set_io_threads
                    "files_identical_to_serial_pass": last_digest == serial_digest}
        # the same pipeline with other writer counts (median of 3 each): is the automatic count the right one on THIS box?
        sweep = {}
        for n in (8, 12, 14, 16, 24):
            device.set_io_threads(n)
            ts = sorted(fresh_pass()[0] for _ in range(3))
            sweep[str(n)] = {"ms": ts[1] * 1e3, "ms_min": ts[0] * 1e3, "ms_max": ts[2] * 1e3}
        device.set_io_threads(0)
        streamed["writer_thread_sweep"] = sweep
        # the same pipeline into the default temporary directory (the box's disk / overlay file system)
        other = tempfile.mkdtemp(prefix="bt_e2e_disk_")
        try:
            dt, _ = streamed_pass(host, LOD_COUNT, other)
            fs_other = "?"
            best = ""
            for ln in open("/proc/mounts"):
                dev, mnt, typ = ln.split()[:3]
                if other.startswith(mnt) and len(mnt) > len(best):
                    best, fs_other = mnt, typ
            streamed["default_tmp_dir"] = {"ms": dt * 1e3, "filesystem": fs_other, "directory": other}
        finally:
            shutil.rmtree(other, ignore_errors=True)
    except Exception as e:
        streamed = {"error": repr(e)}
    files = [f for f in os.listdir(atlas.attachment_directory(root, 0)) if f.endswith(".bin")]
    written = sum(os.path.getsize(os.path.join(atlas.attachment_directory(root, 0), f)) for f in files)
    fs = "?"
    try:
        best = ""
        for line in open("/proc/mounts"):
            dev, mnt, typ = line.split()[:3]
            if root.startswith(mnt) and len(mnt) > len(best):
                best, fs = mnt, typ
    except OSError:
        pass
    # the way back (the streaming side's seam, tile_atlas.rs:77-116 + gpu_tile_atlas.rs:309-336): config.tc + every
    # .bin file -> a fresh atlas, verified against the atlas that wrote them
    load = None
    try:
        atlas2 = bt.TileAtlas.new(cfg, device)
        atlas2.load_tile_config(root)
        device.synchronize()
        t0 = time.perf_counter()
        atlas2.load_tiles(0, root)
        device.synchronize()
        t1 = time.perf_counter()
        order = {(c.side, c.lod, c.x, c.y): i for c, i in atlas.tiles()}
        ok = 0
        for c, i in atlas2.tiles()[:64]:
            ok += int(np.array_equal(atlas2.download_tiles(0, i, 1)[0], atlas.download_tiles(0, order[(c.side, c.lod, c.x, c.y)], 1)[0]))
        load = {"ms": (t1 - t0) * 1e3, "GBps": written / (t1 - t0) / 1e9, "tiles": len(atlas2.tiles()), "spot_checked_identical": ok}
    except Exception as e:  # a side measurement of a side measurement
        load = {"error": repr(e)}
    shutil.rmtree(root, ignore_errors=True)
    up, run, save = results[-1]
    total = up + run + save
    best = streamed["ms"] / 1e3 if streamed and "ms" in streamed else total
    return {"ms": best * 1e3, "tiles_per_s": len(files) / best,
            "pipeline": streamed,  # bt_preprocessor_run_streamed: H2D in bands || kernels || D2H + writes; "ms" above is the median of its 5 passes (ms_min / ms_max: the spread)
            "serial": {"ms": total * 1e3, "tiles_per_s": len(files) / total,
                       "note": "the same span with the legs one after the other: preprocess_tile (upload) -> run -> save"},
            "upload_ms": up * 1e3, "upload_GBps": host.nbytes / up / 1e9,
            "kernels_ms": run * 1e3,
            "save_ms": save * 1e3, "save_GBps": written / save / 1e9, "files": len(files), "bytes_written": written,
            "load_back": load,  # not part of "ms"
            "filesystem": fs, "directory": root, "warm_up_pass_ms": sum(results[0]) * 1e3,
            "span": "source raster in pageable host memory -> hipMalloc + H2D -> 2 kernels -> D2H through 3 pinned "
                    "buffers + writer threads -> 1365 .bin files + config.tc (preprocessor.rs:363,419)"}


def closure_record(device, reps=20, rounds=3):
    """fused_main's two yardsticks measured in THIS process on the context's stream (tools/closure/bt_closure.hip): a linear copy of the
    job's byte mix (0.537 GB read + 0.705 GB written), plain and non-temporal, and the memory skeleton of the kernel (same bytes through
    the same addresses in the same workgroup order, LDS-DMA ring) with 24 packed FMAs per output row and without arithmetic.  Each is the
    best of `rounds` interleaved round averages of `reps` launches; the means stand beside them."""
    import ctypes

    path = os.path.join(ROOT, "tools", "closure", "libbt_closure.so")
    if not os.path.exists(path):
        return {"error": f"{os.path.relpath(path, ROOT)} is not built (python -c 'import __graft_entry__ as g; g.build()')"}
    lib = ctypes.CDLL(path)
    lib.bt_closure_run.restype = ctypes.c_int
    lib.bt_closure_run.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_float), ctypes.POINTER(ctypes.c_uint64)]
    ms = (ctypes.c_float * 8)()
    nbytes = (ctypes.c_uint64 * 2)()
    rc = lib.bt_closure_run(ctypes.c_void_p(device.torch_stream.cuda_stream), reps, rounds, ms, nbytes)
    if rc != 0:
        return {"error": f"bt_closure_run: hipError {rc}"}
    names = ("copy_floor_ms", "copy_floor_nt_ms", "skeleton_ms", "skeleton_no_arith_ms")
    out = {n: float(ms[i]) for i, n in enumerate(names)}
    out.update({n + "_mean": float(ms[4 + i]) for i, n in enumerate(names)})
    out["copy_bytes"] = {"read": int(nbytes[0]), "written": int(nbytes[1])}
    out["method"] = (f"tools/closure: best of {rounds} interleaved round averages of {reps} launches between HIP events on the context's stream, "
                     "in this process, after the timed steps")
    return out


def end_to_end_sharded(args, device, cfg, job, rank, world, collective, src_ptr, dist, fence):
    """N > 1: the reference's span (preprocessor.rs:363,419: sources loaded -> all saves done) with EVERY rank a PCIe link — the one
    configuration that wins by construction.  Distributed result: rank r uploads only its window of the host raster band by band, runs
    its units, writes its finest tiles while later bands run, exchanges the two parent LODs (0.17 GB per rank), runs the finishing
    kernels and writes its share of the lower LODs: together the ranks write the reference's directory, one writer per file.  Span = barrier
    -> every rank's call returned -> barrier, max over ranks; median of 3 passes into fresh directories.  Beside it: the unsharded
    streamed pipeline on rank 0 (the other ranks wait), same invocation."""
    import shutil
    import tempfile

    import numpy as np
    import torch

    import bevy_terrain_amd as bt
    from bevy_terrain_amd.shard import ShardedPreprocess

    host = device.download(src_ptr, (SIZE, SIZE), np.uint16)  # this rank's window of the source (zero = no data elsewhere), in host memory
    parent = ram_directory() or tempfile.gettempdir()

    def shared_dir():
        box = [tempfile.mkdtemp(prefix="bt_e2e_sharded_", dir=parent) if rank == 0 else None]
        dist.broadcast_object_list(box, src=0)
        return box[0]

    def one_pass():
        root = shared_dir()
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new()
        if rank == 0:
            pre.clear_attachment(0, atlas, root)
        j = ShardedPreprocess(pre, atlas, bt.AssetServer().insert("host", host), "host", range(0, LOD_COUNT), rank, world, collective=collective,
                              result="distributed", comm=(job._comm if collective == "library" else None), defer_upload=True)
        fence()
        t0 = time.perf_counter()
        st = j.run_streamed(root)
        device.synchronize()
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        t = torch.tensor([dt], dtype=torch.float64)
        if world > 1:
            t = t.cuda() if dist.get_backend() == "nccl" else t
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        files = len(os.listdir(atlas.attachment_directory(root, 0))) if rank == 0 else 0
        has_tc = os.path.exists(os.path.join(root, cfg.path, "config.tc")) if rank == 0 else False
        fence()
        if rank == 0:
            shutil.rmtree(root, ignore_errors=True)
        pre.close()
        atlas.close()
        return float(t.item()), st, files, has_tc

    one_pass()  # warm: pinned staging buffers, side streams, writer threads' paths
    passes = [one_pass() for _ in range(3)]
    times = sorted(p[0] for p in passes)
    st, files, has_tc = passes[-1][1], passes[-1][2], passes[-1][3]
    per_rank = [None] * world
    dist.all_gather_object(per_rank, {"rank": rank, "uploaded_bytes": st["uploaded_bytes"], "saved_bytes": st["saved_bytes"], "early_tiles": st["early_tiles"],
                                      "bands": st["bands"], "streamed": st["streamed"]})
    out = {"ms": times[1] * 1e3, "ms_min": times[0] * 1e3, "ms_max": times[2] * 1e3, "tiles_per_s": files / times[1] if files else None, "files": files,
           "config_tc": has_tc, "per_rank": per_rank, "result": "distributed", "collective": collective,
           "span": "barrier -> every rank: deferred window upload in bands || kernels || its finest tiles D2H + written; exchange of the two parent LODs; "
                   "finishing kernels; its share of the lower LODs written -> barrier (max over ranks)"}
    # the N = 1 end-to-end span of the same invocation: the unsharded streamed pipeline on rank 0's GPU while the others wait
    n1 = None
    if rank == 0:
        try:
            full_ptr = device.synth_fbm_r16(SIZE, SIZE, SEED)
            full = device.download(full_ptr, (SIZE, SIZE), np.uint16)
            device.free(full_ptr)
            ts = []
            for _ in range(4):
                root = tempfile.mkdtemp(prefix="bt_e2e_n1_", dir=parent)
                atlas = bt.TileAtlas.new(cfg, device)
                pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
                t0 = time.perf_counter()
                pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="host", lod_range=range(0, LOD_COUNT)), bt.AssetServer().insert("host", full), atlas,
                                    defer_upload=True)
                pre.run_streamed(atlas, root)
                ts.append(time.perf_counter() - t0)
                shutil.rmtree(root, ignore_errors=True)
                pre.close()
                atlas.close()
            ts = sorted(ts[1:])
            n1 = {"ms": ts[1] * 1e3, "tiles_per_s": 1365 / ts[1]}
        except Exception as e:
            n1 = {"error": repr(e)}
    fence()
    out["n1_same_invocation"] = n1
    if n1 and "ms" in n1:
        out["speedup_vs_n1_same_invocation"] = n1["ms"] / out["ms"]
    return out


def verify_against(atlas, oracle, shape, held=None):
    """Byte-compare the GPU atlas with the oracle's (same coordinates at the same atlas indices).  held: the atlas layers
    this rank is expected to hold (distributed result), default all."""
    import numpy as np

    sample, lods = shape
    if sample != SIZE:
        return None  # the oracle ran on a window: tile contents differ by construction
    ours = [((c.side, c.lod, c.x, c.y), i) for c, i in atlas.tiles()]
    theirs = oracle.tiles()
    if ours != theirs:
        return {"tiles": len(theirs), "identical": 0, "index_contract": False}
    identical = checked = 0
    for first in range(0, len(theirs), 128):
        count = min(128, len(theirs) - first)
        data = atlas.download_tiles(0, first, count)
        for k in range(count):
            if held is None or (first + k) in held:
                checked += 1
                identical += int(np.array_equal(data[k], oracle.tile(0, first + k)))
    return {"tiles": checked, "identical": identical, "index_contract": True}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="untimed spin-up before the W warm-up steps: the GPU needs ~100 ms of load to reach its sustained clock")
    ap.add_argument("--generic", action="store_true", help="force the reference-shaped batched kernels")
    ap.add_argument("--config", choices=["planar16k", "cube"], default="planar16k",
                    help="planar16k: BASELINE configs 3 / 4, the headline metric (default); cube: config 5's shape, six "
                         "8192^2 faces, lod_count 5, 2046 tiles (a parity / scaling case, not the headline)")
    ap.add_argument("--collective", choices=["library", "torch"], default=None,
                    help="N > 1: who issues the exchange — the library's own RCCL communicator, one grouped collective per "
                         "step (default with the nccl backend), or torch.distributed (gloo test hook)")
    ap.add_argument("--result", choices=["replicated", "distributed"], default="replicated",
                    help="N > 1, the headline pass: replicated (default) = every rank ends with the full atlas (BASELINE config 4); "
                         "distributed = the finest LOD stays on the rank that computed it, only the two parent LODs are exchanged "
                         "(a quarter of the bytes), every rank still holds every lower LOD.  The other one is timed as an extra.")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="N = 1: independent jobs in flight in the HEADLINE pass — P contexts (HIP streams) with an atlas each over "
                         "the same resident source, steps issued round-robin (default 1: one job on one stream, the definition "
                         "that also holds at N > 1, which always uses 1)")
    ap.add_argument("--preflight-timeout", type=int, default=120, help="N > 1: seconds the one-tile-per-rank collective before the timed steps may take")
    ap.add_argument("--extras-timeout", type=int, default=180, help="N > 1: seconds the extra passes may take before the line is printed without them")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the extra passes (N = 1: two jobs in flight; N > 1: kernels only, collective only, the other result mode)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host raster -> files on disk measurement")
    ap.add_argument("--no-workloads", action="store_true",
                    help="N = 1: skip config.workloads (config 2 height / albedo, the masked 16k job re-run and fresh, config 5 height: each timed by events in this process)")
    ap.add_argument("--verify", action="store_true", help="byte-compare all tiles with the checker's run of the same job (oracle/_ref when built)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)
        import socket
        import subprocess

        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    # test hooks (tests/test_shard.py runs the N = 2 path with two processes on ONE GPU, which RCCL refuses):
    # BT_BENCH_DEVICE pins every rank to one device, BT_BENCH_BACKEND=gloo replaces RCCL
    if "BT_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["BT_BENCH_DEVICE"])
    backend = os.environ.get("BT_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    if world > 1:
        import torch.distributed as dist

        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    ranks_seen = 1
    if world > 1:  # every rank answers: the line reports how many did
        ones = torch.ones(1, device="cuda") if backend == "nccl" else torch.ones(1)
        dist.all_reduce(ones)
        ranks_seen = int(ones.item())
        assert ranks_seen == world, f"{ranks_seen} ranks answered, expected {world}"

    import bevy_terrain_amd as bt

    device = bt.Device(local_rank)
    cube = args.config == "cube"
    collective = args.collective or ("library" if backend == "nccl" else "torch")
    result = "replicated" if cube else args.result
    if cube:
        size, lod_count, paths = 8192, 5, [f"synthetic/face{f}" for f in range(6)]
        faces = [device.synth_fbm_r16(size, size, 7 + f) for f in range(6)]
        src_ptr = faces[0]
        cfg = bt.TerrainConfig(lod_count=lod_count, atlas_size=ATLAS_SIZE, path="terrains/bench_cube",
                               model=bt.TerrainModel.sphere((0.0, 0.0, 0.0), 6371000.0, -12000.0, 9000.0))
    else:
        size, lod_count, paths = SIZE, LOD_COUNT, "synthetic/fbm16k"
        if world > 1:  # a rank generates only the window of the source its own launches read (below, once the plan is known)
            src_zero = torch.zeros(SIZE * SIZE * 2, dtype=torch.uint8, device="cuda")
            src_ptr = src_zero.data_ptr()
        else:
            src_ptr = device.synth_fbm_r16(SIZE, SIZE, SEED)
        cfg = bt.TerrainConfig(lod_count=LOD_COUNT, atlas_size=ATLAS_SIZE, path="terrains/bench16k",
                               model=bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=TEXTURE_SIZE, border_size=BORDER,
                                           format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    if cube:
        for p, f in zip(paths, faces):
            server.insert(p, (f, size, size))
    else:
        server.insert(paths, (src_ptr, SIZE, SIZE))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas)
    if world > 1:
        from bevy_terrain_amd.shard import ShardedPreprocess

        # PREFLIGHT, before anything is timed: one grouped in-place all-gather (+ broadcast) of ONE TILE PER RANK through the communicator the
        # steps will use, every byte checked on the host.  A communicator that cannot move a tile fails here, loudly, with RCCL's error
        # string — and a collective that hangs is ended by this watchdog, not by the driver's timeout around the whole run.
        import ctypes
        import threading

        tile_bytes = TEXTURE_SIZE * TEXTURE_SIZE * 2

        def preflight_hung():
            print(f"[rank {rank}] bench.py: the RCCL preflight (one {tile_bytes}-byte tile per rank, {world} ranks) did not complete within "
                  f"{args.preflight_timeout} s — the communicator hangs; no step was timed", file=sys.stderr, flush=True)
            os._exit(3)

        preflight_watchdog = threading.Timer(args.preflight_timeout, preflight_hung)
        preflight_watchdog.daemon = True
        preflight_watchdog.start()
        if collective == "library":
            # every rank must end up on the same path: agree on whether the library's communicator came up everywhere
            ok = 1
            try:
                job = ShardedPreprocess(pre, atlas, server, paths, range(0, lod_count), rank, world, generic=args.generic, collective="library", result=result)
                ms_pre = ctypes.c_float()
                bt._ffi.check(bt._ffi.lib().bt_comm_preflight(job._comm, tile_bytes, ctypes.byref(ms_pre)))
                preflight = {"through": "the library's RCCL communicator (bt_comm_preflight)", "slot_bytes": tile_bytes, "ranks": world,
                             "collective_ms_rank0": ms_pre.value, "verified": "every byte of every rank's slot, on the host"}
            except Exception as e:
                print(f"[rank {rank}] library-issued collective unavailable or failed its preflight ({e!r}); falling back to torch.distributed", file=sys.stderr, flush=True)
                ok = 0
            flag = torch.tensor([ok], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                collective = "torch"
                pre = bt.Preprocessor.new().clear_attachment(0, atlas)
        if collective == "torch":
            job = ShardedPreprocess(pre, atlas, server, paths, range(0, lod_count), rank, world, generic=args.generic, collective="torch", result=result)
            # the same preflight through torch.distributed (backend nccl = RCCL, or gloo in the tests): an exception here ends the run
            slots = torch.zeros(world * tile_bytes, dtype=torch.uint8, device="cuda")
            slots[rank * tile_bytes:(rank + 1) * tile_bytes] = 0x40 + rank
            t_pre = time.perf_counter()
            dist.all_gather_into_tensor(slots, slots[rank * tile_bytes:(rank + 1) * tile_bytes].clone())
            torch.cuda.synchronize()
            got = slots.view(world, tile_bytes).cpu()
            for r in range(world):
                assert bool((got[r] == 0x40 + r).all()), f"[rank {rank}] preflight: slot {r} of the all-gather does not hold rank {r}'s tile"
            preflight = {"through": f"torch.distributed ({backend})", "slot_bytes": tile_bytes, "ranks": world,
                         "collective_ms_rank0": (time.perf_counter() - t_pre) * 1e3, "verified": "every byte of every rank's slot, on the host"}
            del slots
        preflight_watchdog.cancel()
    else:
        if cube:
            pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lod_count)), server, atlas)
        else:
            pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path=paths, lod_range=range(0, lod_count)), server, atlas)
        job = None
    source_window = None
    if world > 1 and not cube:
        # a rank holds its strip of the source + halo.  The plan names the window (bt_preprocessor_source_window);
        # only that part of the synthetic raster is generated on this rank, the rest of the buffer stays zero (= no data: a
        # launch that strayed outside would show in --verify)
        (wx0, wy0, wx1, wy1), _ = job.pre.source_window(atlas, 0, generic=args.generic)
        device.synth_fbm_r16(wx1 - wx0, wy1 - wy0, SEED, x0=wx0, y0=wy0, base_cell=SIZE // 4, dst=src_ptr + (wy0 * SIZE + wx0) * 2, pitch=SIZE * 2)
        device.synchronize()
        source_window = {"x0": wx0, "y0": wy0, "x1": wx1, "y1": wy1, "bytes": (wx1 - wx0) * (wy1 - wy0) * 2, "of_bytes": SIZE * SIZE * 2}

    # N = 1: independent jobs in flight — `depth` contexts (a HIP stream each) with their own atlas, the same queue for
    # each, the source shared in HBM; steps are issued round-robin, so the short serial tail of one job and the drain of
    # its main kernel run beside the main kernel of the next.  (Deferring only the tail to a second stream while the main
    # kernels stay in order was measured and is slower: the tail's workgroups displace persistent main workgroups.)
    depth = max(1, args.pipeline) if job is None else 1
    extras = not args.no_extras
    lanes = [(device, atlas, pre)]
    for _ in range(1, max(depth, 2 if (job is None and extras) else 1)):
        d = bt.Device(local_rank)
        a = bt.TileAtlas.new(cfg, d)
        q = bt.Preprocessor.new().clear_attachment(0, a)
        if cube:
            q.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lod_count)), server, a)
        else:
            q.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path=paths, lod_range=range(0, lod_count)), server, a)
        lanes.append((d, a, q))
    issued = [0]

    def step(profile=False, lane=None, width=None):
        if job is not None:
            job.step(profile)
            return
        _, a, q = lanes[issued[0] % (width or depth) if lane is None else lane]
        issued[0] += 1
        q.run(a, generic=args.generic, keep_queue=True, sync=False, profile=profile)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # the context launches on device.torch_stream; the events are recorded on that same stream
    stream = device.torch_stream
    spin_end = time.perf_counter() + args.spinup_ms / 1e3
    while True:
        go = time.perf_counter() < spin_end
        if world > 1:  # every rank must run the same number of steps (each one ends in collectives): rank 0 decides
            flag = torch.tensor([1 if go else 0], device="cuda")
            dist.broadcast(flag, src=0)
            go = bool(flag.item())
        if not go:
            break
        for _ in range(16):
            step()
        fence()
    for _ in range(args.warmup):
        step()
    fence()
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    stride = max(1, min(8, args.steps // 5))  # instrumented steps: every 8th of the default 200, at least five of a short run
    t0 = time.perf_counter()
    start.record(stream)
    for i in range(args.steps):
        # one stream: per-launch HIP events (the roofline's launch durations) on every `stride`-th step — each event costs ~2.6 us of
        # stream time, an instrumented step ~10 us more than a plain one; several jobs in flight: none here — a launch that shares the
        # GPU with another job's says nothing about the kernel, the durations come from the one-stream pass below
        step(profile=depth == 1 and (i % stride == 0) and not os.environ.get("BT_BENCH_NO_LAUNCH_EVENTS"))
    stops = []
    for d, _, _ in lanes[:depth]:  # the K steps are over when the last lane's stream has drained
        e = torch.cuda.Event(enable_timing=True)
        e.record(d.torch_stream)
        stops.append(e)
    fence()
    wall_ms = (time.perf_counter() - t0) * 1e3
    ms = max(start.elapsed_time(e) for e in stops)
    if world > 1:
        t = torch.tensor([ms], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms = float(t.item())
    ms_per_step = ms / args.steps

    # N > 1: the N = 1 step of the SAME invocation — rank 0 runs the whole, unsharded job once more on its GPU (its own context, atlas and a
    # complete source) while the other ranks wait at a barrier, so that a SCALE record carries its own baseline: same box, same clocks, same build
    n1_reference = None
    if world > 1 and not cube and not args.no_extras:
        if rank == 0:
            try:
                d1 = bt.Device(local_rank)
                full = d1.synth_fbm_r16(SIZE, SIZE, SEED)
                a1 = bt.TileAtlas.new(cfg, d1)
                q1 = bt.Preprocessor.new().clear_attachment(0, a1).preprocess_tile(
                    bt.PreprocessDataset(attachment_index=0, path="n1", lod_range=range(0, lod_count)), bt.AssetServer().insert("n1", (full, SIZE, SIZE)), a1)
                for _ in range(max(10, args.warmup)):
                    q1.run(a1, generic=args.generic, keep_queue=True, sync=False)
                d1.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(d1.torch_stream)
                for _ in range(args.steps):
                    q1.run(a1, generic=args.generic, keep_queue=True, sync=False)
                e1.record(d1.torch_stream)
                d1.synchronize()
                t1 = e0.elapsed_time(e1) / args.steps
                n1_reference = {"ms_per_step": t1, "tiles_per_s": q1.stats()["tiles"] / (t1 / 1e3),
                                "note": "the unsharded job on rank 0's GPU, one stream, K steps between HIP events, measured in this invocation while the other ranks wait"}
                q1.close()
                d1.free(full)
                del a1
            except Exception as e:  # never lose the headline over the reference
                n1_reference = {"error": repr(e)}
        fence()

    # N > 1: every rank preprocesses a WHOLE 16k job of its own at the same time — no strip, no collective: what a host with one terrain per GPU
    # gets ("independent terrain tiles shard naturally"); aggregate tiles/s = N x tiles / the slowest rank's step.  Not the headline (that is ONE
    # job shared by the ranks); reported so that the line holds the collective-free end of the range next to the replicated one.
    independent = None
    if world > 1 and not cube and not args.no_extras:
        try:
            d2 = bt.Device(local_rank)
            full2 = d2.synth_fbm_r16(SIZE, SIZE, SEED + rank)
            a2 = bt.TileAtlas.new(cfg, d2)
            q2 = bt.Preprocessor.new().clear_attachment(0, a2).preprocess_tile(
                bt.PreprocessDataset(attachment_index=0, path="own", lod_range=range(0, lod_count)), bt.AssetServer().insert("own", (full2, SIZE, SIZE)), a2)
            for _ in range(max(10, args.warmup)):
                q2.run(a2, generic=args.generic, keep_queue=True, sync=False)
            d2.synchronize()
            fence()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(d2.torch_stream)
            for _ in range(args.steps):
                q2.run(a2, generic=args.generic, keep_queue=True, sync=False)
            e1.record(d2.torch_stream)
            d2.synchronize()
            t2 = torch.tensor([e0.elapsed_time(e1) / args.steps], device="cuda")
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
            t2 = float(t2.item())
            independent = {"ms_per_step": t2, "tiles_per_s": world * q2.stats()["tiles"] / (t2 / 1e3), "jobs": world,
                           "note": "one whole 16k job per rank at the same time (seeds 42 + rank), no collective; max over ranks"}
            q2.close()
            d2.free(full2)
            del a2
        except Exception as e:
            independent = {"error": repr(e)}
        fence()

    def timed_pass(fn, on_stream=None):
        """K calls of fn between two events on `on_stream`, fenced on both sides, max over ranks; ms per call"""
        st = on_stream or stream
        fence()
        start.record(st)
        for i in range(args.steps):
            fn(i)
        stop.record(st)
        fence()
        t = start.elapsed_time(stop)
        if world > 1:
            tt = torch.tensor([t], device="cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = float(tt.item())
        return t / args.steps

    compute_only_ms = exchange_only_ms = None
    other_result = None
    overlapped = None       # N > 1 extra: {"ms_per_step", "tiles_per_s"} of the overlapped pair (two atlases per rank)
    overlap_atlases = []    # ... and its atlases, for --verify

    # N = 1 extra: two independent jobs in flight (two contexts / streams, an atlas each): the tail, the todo launch and the
    # drain of one job's main kernel run beside the main kernel of the next
    two_in_flight_ms = None
    if job is None and depth == 1 and extras and len(lanes) >= 2:
        for _ in range(8):
            step(width=2)
        fence()
        start.record(stream)
        for _ in range(args.steps):
            step(width=2)
        ends = []
        for d, _, _ in lanes[:2]:
            e = torch.cuda.Event(enable_timing=True)
            e.record(d.torch_stream)
            ends.append(e)
        fence()
        two_in_flight_ms = max(start.elapsed_time(e) for e in ends) / args.steps

    # N = 1: fused_main's yardsticks (linear copy of the byte mix, memory skeleton) in this process, right behind the timed steps
    closure = None
    if job is None and world == 1 and extras and not cube and not args.generic:
        try:
            closure = closure_record(device)
        except Exception as e:  # never lose the headline over a side measurement
            closure = {"error": repr(e)}

    # N = 1 with several jobs in flight: the same K steps once more on ONE stream — the step time without overlap, and the
    # undisturbed per-launch durations the roofline is computed from (HIP events on that stream, every `stride`-th step)
    one_stream_ms = None
    if job is None and depth > 1:
        fence()
        start.record(stream)
        for i in range(args.steps):
            step(profile=(i % stride == 0) and not os.environ.get("BT_BENCH_NO_LAUNCH_EVENTS"), lane=0)
        stop.record(stream)
        fence()
        one_stream_ms = start.elapsed_time(stop) / args.steps

    stats = pre.stats() if job is None else job.stats()
    tiles = stats["tiles"]
    launches = pre.profile() if job is None else job.profile()
    dominant = max(launches, key=lambda l: l["avg_ms"]) if launches and launches[0]["samples"] else None

    line = {
        "metric": "terrain tiles/sec preprocessed (16k^2 heightmap)",
        "value": tiles / (ms_per_step / 1e3),
        "unit": "tiles/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "strong",  # the 16k job is fixed: N ranks share it (column strips), per-GPU work shrinks with N
        "collective_backend": (backend if collective == "torch" else "rccl (library-issued, one grouped collective per step)") if world > 1 else None,
        "vs_baseline": None,
        "dtype": "f32",  # IEEE binary32 arithmetic on u16 texels, results bit-exact vs the oracle
        "data": "synthetic",
        "config": {"workload": (f"synthetic cube: 6 faces of {size}x{size} fBm R16 (seeds 7..12), T={TEXTURE_SIZE}, b={BORDER}, "
                                f"lod_count={lod_count}: split + pyramid + stitch (cube seams) into {tiles} tiles [BASELINE config 5's shape]"
                                if cube else
                                f"synthetic {SIZE}x{SIZE} fBm R16 heightmap (seed {SEED}), T={TEXTURE_SIZE}, b={BORDER}, "
                                f"lod_count={LOD_COUNT}: split + pyramid + stitch into {tiles} tiles"),
                   "path": "generic (batched split/downsample/stitch)" if stats["fused_jobs"] == 0 else "fused",
                   "jobs_in_flight": depth,  # contexts (streams) + atlases the K steps of the headline rotate over
                   "ranks_seen": ranks_seen,
                   "ms_per_step_one_stream": one_stream_ms if depth > 1 else ms_per_step,
                   "ms_per_step_two_in_flight": two_in_flight_ms,  # N = 1 extra pass: two contexts (streams + atlases), steps round-robin
                   "tiles_per_s_two_in_flight": (tiles / (two_in_flight_ms / 1e3)) if two_in_flight_ms else None,
                   "kernels_per_step": stats["kernel_launches"],  # fused_main, fused_tail: what rocprofv3 --stats counts
                   "algorithmic_bytes_per_step": stats["algorithmic_bytes"],
                   "whole_step_GBps": stats["algorithmic_bytes"] / (ms_per_step / 1e3) / 1e9,
                   "host_wall_ms_per_step": wall_ms / args.steps,
                   "kernels_only_ms_per_step": compute_only_ms,  # N > 1: without the all-gathers
                   "collective_only_ms_per_step": exchange_only_ms,  # N > 1: the grouped collective alone
                   "other_result_mode": other_result,  # N > 1: the same job with the other --result, timed in the same run
                   "all_gather_bytes_per_rank": (job.gather_bytes if job is not None else 0),
                   "rccl_preflight": preflight if world > 1 else None,  # N > 1: the one-tile-per-rank collective that ran before anything was timed
                   "n1_same_invocation": n1_reference,  # N > 1: the unsharded step on rank 0's GPU, this invocation
                   "independent_jobs": independent,  # N > 1: one whole job per rank at the same time, no collective (aggregate tiles/s)
                   "source_window_rank0": source_window,  # N > 1: the part of the source rank 0 generated (its strips + halo); the rest stays zero
                   "result": (None if job is None else
                              "replicated: every rank ends with the full atlas" if job.held is None else
                              "distributed: the finest LOD stays on the rank that computed it (complete there), the two parent "
                              "LODs are exchanged, every rank holds every lower LOD"),
                   "launches": launches},
    }
    # N > 1 extras: the same K steps without the collectives (kernels only), the collectives without the kernels, and the
    # other result mode (SURVEY §8e asks for compute and compute + collective separately).  They come AFTER the headline is
    # complete, under a watchdog: these passes end in collectives, and a rank that fails inside one leaves the others waiting —
    # the headline line must not be lost to that (rank 0 prints what it has, every rank leaves).
    if job is not None and extras:
        import threading

        def bail():
            if rank == 0:
                line["config"]["extras_error"] = f"the extra passes did not finish within {args.extras_timeout} s"
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog = threading.Timer(args.extras_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
        try:
            compute_only_ms = timed_pass(lambda i: job.step(False, gather=False))
            job.step(False)  # a complete atlas again
            exchange_only_ms = timed_pass(lambda i: job.exchange())
            job.step(False)  # leave a complete atlas behind (--verify)
            fence()
            if not cube:
                # the other result mode on a second atlas / queue / communicator of this rank; every rank must agree that it came up
                other = "distributed" if result == "replicated" else "replicated"
                ok, job2 = 1, None
                try:
                    atlas2 = bt.TileAtlas.new(cfg, device)
                    pre2 = bt.Preprocessor.new().clear_attachment(0, atlas2)
                    job2 = ShardedPreprocess(pre2, atlas2, server, paths, range(0, lod_count), rank, world, generic=args.generic, collective=collective, result=other)
                except Exception as e:
                    print(f"[rank {rank}] extra pass ({other}) unavailable: {e!r}", file=sys.stderr)
                    ok = 0
                flag = torch.tensor([ok], device="cuda")
                dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                if int(flag.item()) == 1:
                    for _ in range(3):
                        job2.step(False)
                    t_other = timed_pass(lambda i: job2.step(False))
                    other_result = {"result": other, "ms_per_step": t_other, "tiles_per_s": job2.stats()["tiles"] / (t_other / 1e3),
                                    "all_gather_bytes_per_rank": job2.gather_bytes}
                if job2 is not None:
                    job2.close()
            # the same result mode with the exchange of step k hidden behind the kernels of step k + 1: a second atlas +
            # preprocessor of this rank, the same communicator, collectives on its own queue (bevy_terrain_amd.shard.OverlappedSharded)
            from bevy_terrain_amd.shard import OverlappedSharded

            ok, job_b = 1, None
            try:
                atlas_b = bt.TileAtlas.new(cfg, device)
                pre_b = bt.Preprocessor.new().clear_attachment(0, atlas_b)
                job_b = ShardedPreprocess(pre_b, atlas_b, server, paths, range(0, lod_count), rank, world, generic=args.generic, collective=collective,
                                          result=result, comm=job._comm)
            except Exception as e:
                print(f"[rank {rank}] overlapped pass unavailable: {e!r}", file=sys.stderr)
                ok = 0
            flag = torch.tensor([ok], device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 1:
                pair = OverlappedSharded(job, job_b)
                for _ in range(4):
                    pair.step()
                t_pair = timed_pass(lambda i: pair.step())
                pair.flush()
                fence()
                overlapped = {"ms_per_step": t_pair, "tiles_per_s": tiles / (t_pair / 1e3), "atlases_per_rank": 2,
                              "note": "step k's grouped collective on the communicator's own stream while step k + 1's local kernels run; "
                                      "the finishing kernels of a step follow its collective one step late"}
                overlap_atlases.append(atlas_b)
        except Exception as e:  # (an exception every rank raises at the same point, e.g. an unsupported flag)
            line["config"]["extras_error"] = repr(e)
        finally:
            watchdog.cancel()
        line["config"]["kernels_only_ms_per_step"] = compute_only_ms
        line["config"]["collective_only_ms_per_step"] = exchange_only_ms
        line["config"]["other_result_mode"] = other_result
        line["config"][f"{result}_overlapped"] = overlapped
    if n1_reference and "ms_per_step" in n1_reference:
        # strong scaling against the N = 1 step of this invocation: speedup = t1 / tN, efficiency = speedup / N (informational: the driver
        # computes its own from the per-N lines)
        def ratio(t):
            return {"speedup": n1_reference["ms_per_step"] / t, "efficiency": n1_reference["ms_per_step"] / t / world} if t else None

        sc = {"headline": ratio(ms_per_step)}
        if overlapped:
            sc[f"{result}_overlapped"] = ratio(overlapped["ms_per_step"])
        if other_result:
            sc[other_result["result"]] = ratio(other_result["ms_per_step"])
        if compute_only_ms:
            sc["kernels_only"] = ratio(compute_only_ms)
        line["config"]["scaling_vs_n1_same_invocation"] = sc
    if dominant:
        achieved = dominant["algorithmic_bytes"] / (dominant["avg_ms"] / 1e3) / 1e9
        # HBM bytes per launch from the PMC passes of the same command (rocprofv3 cannot run inside this
        # process): the newest committed summary under profiles/ that knows this kernel, else null
        traffic, traffic_source = None, None
        import glob

        for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_summary.json")), reverse=True):
            try:
                summary = json.load(open(path))
                traffic = summary[dominant["kind"]]["hbm_traffic_bytes"]["total"]
                traffic_source = {"file": os.path.relpath(path, ROOT), "git": summary.get("_git"), "workload": summary.get("_workload"),
                                  "note": "rocprofv3 --pmc passes of this same command (tools/profile_round.sh); PMC collection "
                                          "cannot run inside the timed process"}
                break
            except (KeyError, ValueError, OSError):
                continue
        line["roofline"] = {"bound": "hbm", "kernel": dominant["kind"], "achieved": achieved, "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                            "traffic_source": traffic_source,
                            # (informational: the same guide measures 6.29 TB/s for a float4 copy, "≈6.3 TB/s achievable", MI355X_MICROARCH.md:35,293)
                            "achievable_peak": HBM_ACHIEVABLE_GBS, "frac_of_achievable": achieved / HBM_ACHIEVABLE_GBS,
                            "avg_launch_ms": dominant["avg_ms"], "algorithmic_bytes_per_launch": dominant["algorithmic_bytes"],
                            # the closure record (same process, same stream): a linear copy of the same byte mix and the kernel's memory skeleton
                            "copy_floor_ms": (closure or {}).get("copy_floor_ms"), "skeleton_ms": (closure or {}).get("skeleton_ms"),
                            "frac_of_copy": ((closure["copy_floor_ms"] / dominant["avg_ms"]) if closure and closure.get("copy_floor_ms") and dominant["kind"] == "fused_main" else None),
                            "kernel_over_skeleton": ((dominant["avg_ms"] / closure["skeleton_ms"]) if closure and closure.get("skeleton_ms") and dominant["kind"] == "fused_main" else None),
                            "closure": closure,
                            "launch_timing": (f"HIP events on the launching stream, every {stride}th step of the timed K steps" if depth == 1 else
                                              f"HIP events on the launching stream, every {stride}th step of the one-stream pass of the same K steps "
                                              f"(in the timed pass {depth} jobs share the GPU: a launch's span there is not the kernel's duration)")}
    if cube:
        args.no_cpu_baseline = args.no_end_to_end = True  # the side measurements belong to the headline workload
        args.verify = False
    if rank == 0 and world == 1 and not args.no_end_to_end and not os.environ.get("BT_BENCH_SKIP_16K_E2E"):
        try:
            line["end_to_end"] = end_to_end(device, src_ptr)
        except Exception as e:  # never lose the headline line over a side measurement
            line["end_to_end"] = {"error": repr(e)}
    if rank == 0 and world == 1 and not args.no_workloads and not cube:
        # every other fraction DESIGN.md claims, timed by events in this process (outside the headline's timed steps)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import workloads

            line["config"]["workloads"] = workloads.all_workloads(device) if not os.environ.get("BT_BENCH_SKIP_WORKLOADS") else None
        except Exception as e:
            line["config"]["workloads"] = {"error": repr(e)}
        if not args.no_end_to_end:
            line.setdefault("end_to_end", {})
            # the reference's own two examples end to end (preprocessor.rs:363,419), streamed with the serial legs beside them
            try:
                line["end_to_end"]["reference_examples"] = {
                    "config2_planar_height_albedo": workloads.end_to_end_config2(device, passes=9),  # (7 ms a pass; the first pass behind the serial ones is an outlier)
                    "config5_cube_height": workloads.end_to_end_config5(device, passes=5)}
            except Exception as e:
                line["end_to_end"]["reference_examples"] = {"error": repr(e)}
    if world > 1 and not cube and extras and not args.no_end_to_end:
        import threading

        def bail_e2e():  # these passes end in collectives too: a rank that fails inside one must not cost the line
            if rank == 0:
                line["end_to_end_sharded"] = {"error": f"did not finish within {args.extras_timeout} s"}
                print(json.dumps(line), flush=True)
            os._exit(0)

        watchdog_e2e = threading.Timer(args.extras_timeout, bail_e2e)
        watchdog_e2e.daemon = True
        watchdog_e2e.start()
        try:
            line["end_to_end_sharded"] = end_to_end_sharded(args, device, cfg, job, rank, world, collective, src_ptr, dist, fence)
        except Exception as e:  # (an exception every rank raises at the same point)
            line["end_to_end_sharded"] = {"error": repr(e)}
        finally:
            watchdog_e2e.cancel()
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(device, src_ptr)  # reported at N = 1 only
    if rank == 0 and args.verify:
        full_src = device.synth_fbm_r16(SIZE, SIZE, SEED) if source_window is not None else src_ptr  # the checker works from the whole raster
        oracle, shape, line["verify_checker"] = oracle_atlas(device, full_src)
        if full_src != src_ptr:
            device.free(full_src)
        held = None
        if job is not None and job.held is not None:  # distributed result: rank 0 holds its finest pieces + every lower LOD
            finest = max(c[1] for c, _ in oracle.tiles())
            held = {i for c, i in oracle.tiles() if c[1] < finest}
            for piece in job.held:
                held.update(range(piece["first_layer"], piece["first_layer"] + piece["layers"]))
        line["verify_vs_oracle"] = verify_against(atlas, oracle, shape, held)
        for k, (_, a, _) in enumerate(lanes[1:], 1):  # every lane's atlas holds a complete, identical job
            line[f"verify_vs_oracle_lane{k}"] = verify_against(a, oracle, shape)
        for a in overlap_atlases:  # the second atlas of the overlapped pair: the same job, finished one step late
            line["verify_vs_oracle_overlapped_second_atlas"] = verify_against(a, oracle, shape, held)
    if world > 1:
        dist.barrier()  # the other ranks wait for rank 0's oracle run before tearing the group down
    if rank == 0 and world == 1:
        # the other half of the hot path, for the record: the per-frame tiling prepass on scripted camera paths
        # (latency-bound, one launch per frame; not part of `value`)
        try:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import refine_bench

            line["config"]["tiling_prepass"] = refine_bench.measure(device)
        except Exception as e:  # never lose the headline line over the side measurement
            line["config"]["tiling_prepass"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
