#!/usr/bin/env python3
"""The non-headline workloads whose roofline fractions DESIGN.md quotes, as functions shared by bench.py (`config.workloads`: the driver's
BENCH line then carries every claimed fraction) and tools/config_bench.py (the same numbers under rocprofv3 / from the builder's leases):

  config2_height_4k / config2_albedo_4k   BASELINE config 2 (examples/preprocess_planar.rs): 4096^2 R16 / Rgba8, lod_count 4, 85 tiles each
  config3_masked_16k                      the headline job with the 5 % no-data mask, re-run on a written atlas (previous values are fetched)
  config3_masked_16k_fresh                the same on atlases nothing has written since bt_atlas_create (both reference examples: clear_attachment,
                                          then one dataset) — FusedArgs::prev_zero, no previous-value fetches
  config5_cube_height_8k                  BASELINE config 5's height attachment: 6 faces of 8192^2 R16, lod_count 5, 2046 tiles
  config5_cube_albedo_8k                  ... and its albedo attachment: 6 faces of 8192^2 Rgba8, 2046 tiles of 1 MiB

and the end-to-end span (preprocessor.rs:363,419: sources loaded -> all saves done) of the reference's two examples through
bt_preprocessor_run_streamed, with the serial legs beside it.  Every job is timed between HIP events on the context's stream."""
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bevy_terrain_amd as bt  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md


def launches_of(prof):
    return [(l["kind"], round(l["avg_ms"] * 1e3, 1), l["algorithmic_bytes"]) for l in prof]


def entry(ms, prof, st, **more):
    out = {"ms": ms, "tiles": st["tiles"], "algorithmic_bytes": st["algorithmic_bytes"], "GBps": st["algorithmic_bytes"] / ms / 1e6,
           "frac": st["algorithmic_bytes"] / ms / 1e6 / HBM_PEAK_GBS, "launches": launches_of(prof) if prof else None,
           "prev_zero_launches": st.get("prev_zero_launches")}
    if st.get("ms_with_launch_events") is not None:  # the pass the `launches` come from (each launch between two events: slower, see time_job)
        out["ms_with_launch_events"] = st["ms_with_launch_events"]
    out.update(more)
    return out


SPINUP_MS = 150.0  # the GPU needs ~100 ms of load to reach its sustained clock (bench.py --spinup-ms): the cube job measured 0.466 ms behind 10 warm-up runs, 0.437 behind this


def spin_up(device, pre, atlas, ms=None):
    """untimed re-runs of the kept queue for `ms` of wall time"""
    end = time.perf_counter() + (SPINUP_MS if ms is None else ms) / 1e3
    while time.perf_counter() < end:
        for _ in range(16):
            pre.run(atlas, keep_queue=True, sync=False)
        device.synchronize()


def time_job(device, pre, atlas, steps=50, warm=10):
    """`steps` re-runs of a kept queue between two events, behind a spin-up — and NOTHING else in the stream: the per-launch events of
    BT_RUN_PROFILE cost the stream ~2.6 us each (a 2-launch job: + 10 us per run; config 2's height job 27.7 -> 38.3 us, found at the end of
    round 6 — rounds 2 - 6 quoted the instrumented time).  The launch durations come from a second, instrumented pass."""
    if warm > 2:  # (warm <= 2: counter passes under rocprofv3 — every launch is collected, clocks do not matter)
        spin_up(device, pre, atlas)
    for _ in range(warm):
        pre.run(atlas, keep_queue=True, sync=False)
    device.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(device.torch_stream)
    for _ in range(steps):
        pre.run(atlas, keep_queue=True, sync=False)
    e.record(device.torch_stream)
    device.synchronize()
    ms = s.elapsed_time(e) / steps
    s.record(device.torch_stream)
    for _ in range(min(steps, 50)):
        pre.run(atlas, keep_queue=True, sync=False, profile=True)
    e.record(device.torch_stream)
    device.synchronize()
    st = dict(pre.stats())
    st["ms_with_launch_events"] = s.elapsed_time(e) / min(steps, 50)
    return ms, pre.profile(), st


def planar_cfg(lods, atlas_size, path, attachments):
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=atlas_size, path=path, model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    for name, fmt in attachments:
        cfg.add_attachment(bt.AttachmentConfig(name=name, texture_size=512, border_size=2, format=fmt))
    return cfg


def config2_sources(device):
    h = device.synth_fbm_r16(4096, 4096, 1234)
    rng = np.random.default_rng(1235)
    albedo = rng.integers(1, 256, size=(4096, 4096, 4), dtype=np.uint8)
    return h, albedo


def config2(device, steps=50, only=None):
    h, albedo = config2_sources(device)
    cfg = planar_cfg(4, 1024, "terrains/planar", [("height", bt.AttachmentFormat.R16), ("albedo", bt.AttachmentFormat.Rgba8)])
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("h", (h, 4096, 4096)).insert("a", albedo)
    out = {}
    for name, att, path in (("config2_height_4k", 0, "h"), ("config2_albedo_4k", 1, "a")):
        if only and only != name:
            continue
        pre = bt.Preprocessor.new().clear_attachment(att, atlas).preprocess_tile(
            bt.PreprocessDataset(attachment_index=att, path=path, lod_range=range(0, 4)), server, atlas)
        ms, prof, st = time_job(device, pre, atlas, steps)
        out[name] = entry(ms, prof, st)
        pre.close()
    device.free(h)
    return out


def masked_source(device, size=16384, mask=True):
    """the 16k fBm raster with the 5 % no-data mask of tests/test_gpu_preprocess.py (seed 43: 37 x 53 texel cells + single texels)"""
    ptr = device.synth_fbm_r16(size, size, 42)
    if not mask:
        return ptr
    src = device.download(ptr, (size, size), np.uint16)
    rng = np.random.default_rng(43)
    cells = rng.random((size // 37 + 1, size // 53 + 1)) < 0.05
    mask = np.repeat(np.repeat(cells, 37, axis=0), 53, axis=1)[:size, :size]
    single = rng.integers(0, size, size=(size, 2))
    mask[single[:, 0], single[:, 1]] = True
    src[mask] = 0
    device.free(ptr)
    return device.upload(src)


def masked16k(device, steps=20, fresh_atlases=7, rerun=True, mask=True):
    """-> {"config3_masked_16k": re-run on a written atlas, "config3_masked_16k_fresh": each run on an atlas nothing has written}"""
    size, lods = 16384, 6
    ptr = masked_source(device, size, mask)
    cfg = planar_cfg(lods, 2048, "terrains/masked16k", [("height", bt.AttachmentFormat.R16)])
    server = bt.AssetServer().insert("m", (ptr, size, size))
    ds = bt.PreprocessDataset(attachment_index=0, path="m", lod_range=range(0, lods))
    out = {}
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(ds, server, atlas)
    if rerun:  # (rerun=False: counter passes of the fresh runs alone — no launch of the fetching kind in the process)
        ms, prof, st = time_job(device, pre, atlas, steps)
        out["config3_masked_16k"] = entry(ms, prof, st, note="re-runs of a kept queue: every tile is written, previous values are fetched")
    if fresh_atlases:
        jobs = []
        for _ in range(fresh_atlases):
            a = bt.TileAtlas.new(cfg, device)
            q = bt.Preprocessor.new().clear_attachment(0, a).preprocess_tile(ds, server, a)
            q.source_window(a, 0)  # compiles the plan (host work outside the timed span)
            jobs.append((a, q))
        device.synchronize()
        if rerun:  # clocks up, on the written atlas
            spin_up(device, pre, atlas)
        events = [torch.cuda.Event(enable_timing=True) for _ in range(fresh_atlases + 1)]
        flagged = []
        events[0].record(device.torch_stream)
        for k, (a, q) in enumerate(jobs):  # (the last run carries the per-launch events the `launches` come from and is not among the timed ones)
            q.run(a, keep_queue=True, sync=False, profile=k == fresh_atlases - 1)
            events[k + 1].record(device.torch_stream)
            flagged.append(q.stats()["prev_zero_launches"])
        device.synchronize()
        times = sorted(events[k].elapsed_time(events[k + 1]) for k in range(max(fresh_atlases - 1, 1)))
        prof = jobs[-1][1].profile()
        st = dict(jobs[-1][1].stats())
        st["prev_zero_launches"] = min(flagged)
        out["config3_masked_16k_fresh"] = entry(times[len(times) // 2], prof, st, ms_min=times[0], ms_max=times[-1], runs=len(times),
                                                 note="each run on an atlas nothing has written since bt_atlas_create (median of the runs; launches: the last run's)")
        for a, q in jobs:
            q.close()
    pre.close()
    device.free(ptr)
    return out


def cube_faces(device, size=8192, fmt="r16"):
    return [(device.synth_fbm_r16(size, size, 7 + s), size, size) for s in range(6)]


def config5_height(device, steps=50):
    faces = cube_faces(device)
    cfg = bt.TerrainConfig(lod_count=5, atlas_size=2048, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    paths = [f"face{s}" for s in range(6)]
    for p, f in zip(paths, faces):
        server.insert(p, f)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, 5)), server, atlas)
    ms, prof, st = time_job(device, pre, atlas, steps)
    pre.close()
    for f, _, _ in faces:
        device.free(f)
    return {"config5_cube_height_8k": entry(ms, prof, st)}


def config5_albedo(device, steps=20):
    """BASELINE config 5's second attachment: 6 faces of 8192^2 Rgba8 (1.6 GB of source), lod_count 5, 2046 tiles of 1 MiB"""
    rng = np.random.default_rng(77)
    face = rng.integers(1, 256, size=(8192, 8192, 4), dtype=np.uint8)
    cfg = bt.TerrainConfig(lod_count=5, atlas_size=2048, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=512, border_size=2, format=bt.AttachmentFormat.Rgba8))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    paths = [f"albedo{s}" for s in range(6)]
    for i, p in enumerate(paths):
        server.insert(p, np.roll(face, 997 * i, axis=1))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, 5)), server, atlas)
    ms, prof, st = time_job(device, pre, atlas, steps)
    pre.close()
    return {"config5_cube_albedo_8k": entry(ms, prof, st)}


def all_workloads(device, steps=None):
    """the five entries of bench.py's `config.workloads` (each job < 1.2 ms; about 15 s in all, most of it building the masked raster)"""
    out = {}
    out.update(config2(device, steps or 50))
    out.update(masked16k(device, steps or 20))
    out.update(config5_height(device, steps or 30))
    out.update(config5_albedo(device, steps or 20))
    return out


# ------------------------------------------------------------------------------------------------ end to end: the reference's two examples

def _ram_directory():
    try:
        st = os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize > (6 << 30):
            return "/dev/shm"
    except OSError:
        pass
    return None


def _median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def end_to_end_example(device, cfg, queue_jobs, passes=5):
    """One of the reference's examples end to end: `queue_jobs(pre, atlas, defer)` queues its datasets from HOST rasters (deferred for the
    streamed pipeline).  Returns streamed (median of `passes` passes into fresh directories) and serial (upload, run, save one after the
    other) spans, the bytes moved, and max(upload leg, download + write leg) measured alone — the pipeline's floor on this host."""
    parent = _ram_directory()
    n_att = len(cfg.attachments)

    def one(streamed):
        root = tempfile.mkdtemp(prefix="bt_e2e_ex_", dir=parent)
        try:
            atlas = bt.TileAtlas.new(cfg, device)
            device.synchronize()
            pre = bt.Preprocessor.new()
            for ai in range(n_att):
                pre.clear_attachment(ai, atlas, root)
            t0 = time.perf_counter()
            queue_jobs(pre, atlas, streamed)
            t1 = time.perf_counter()
            if streamed:
                st = pre.run_streamed(atlas, root)
                t2 = t3 = time.perf_counter()
            else:
                pre.run(atlas)
                t2 = time.perf_counter()
                pre.save(atlas, root)
                t3 = time.perf_counter()
                st = None
            files = sum(len(os.listdir(atlas.attachment_directory(root, ai))) for ai in range(n_att))
            pre.close()
            atlas.close()
            return {"ms": (t3 - t0) * 1e3, "upload_ms": (t1 - t0) * 1e3, "kernels_ms": (t2 - t1) * 1e3, "save_ms": (t3 - t2) * 1e3, "stats": st, "files": files}
        finally:
            shutil.rmtree(root, ignore_errors=True)

    one(False)  # warm: staging buffers, side streams, writer threads' paths
    one(True)
    serial = [one(False) for _ in range(3)]
    streamed = [one(True) for _ in range(passes)]
    st = streamed[-1]["stats"]
    s_med = _median([r["ms"] for r in serial])
    best_serial = min(serial, key=lambda r: r["ms"])
    floor = max(_median([r["upload_ms"] for r in serial]), _median([r["save_ms"] for r in serial]))
    t = sorted(r["ms"] for r in streamed)
    return {"ms": t[len(t) // 2], "ms_min": t[0], "ms_max": t[-1], "ms_all": [round(r["ms"], 3) for r in streamed], "passes": passes, "files": streamed[-1]["files"],
            "streamed": bool(st["streamed"]), "bands": st["bands"], "banded_launches": st["banded_launches"], "early_tiles": st["early_tiles"],
            "uploaded_bytes": st["uploaded_bytes"], "saved_bytes": st["saved_bytes"],
            "serial": {"ms": s_med, "ms_all": [round(r["ms"], 3) for r in serial], "upload_ms": best_serial["upload_ms"], "kernels_ms": best_serial["kernels_ms"], "save_ms": best_serial["save_ms"]},
            "max_leg_ms": floor, "over_max_leg": t[len(t) // 2] / floor if floor else None,
            "upload_GBps": st["uploaded_bytes"] / floor / 1e6 if floor else None, "filesystem_parent": parent or tempfile.gettempdir()}


def end_to_end_config2(device, passes=9):
    """examples/preprocess_planar.rs:16-60: height (R16) + albedo (Rgba8), 4096^2 each, lod_count 4 -> 2 x 85 tiles"""
    h_dev, albedo = config2_sources(device)
    height = device.download(h_dev, (4096, 4096), np.uint16)
    device.free(h_dev)
    cfg = planar_cfg(4, 1024, "terrains/planar", [("height", bt.AttachmentFormat.R16), ("albedo", bt.AttachmentFormat.Rgba8)])
    server = bt.AssetServer().insert("h", height).insert("a", albedo)

    def queue_jobs(pre, atlas, defer):
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, 4)), server, atlas, defer_upload=defer)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=1, path="a", lod_range=range(0, 4)), server, atlas, defer_upload=defer)

    return end_to_end_example(device, cfg, queue_jobs, passes)


def end_to_end_config5(device, passes=7):
    """examples/preprocess_spherical.rs:20-48 (height attachment): six faces of 8192^2 R16, lod_count 5 -> 2046 tiles (0.8 GB in, 1.07 GB out)"""
    faces = []
    for ptr, w, h in cube_faces(device):
        faces.append(device.download(ptr, (h, w), np.uint16))
        device.free(ptr)
    cfg = bt.TerrainConfig(lod_count=5, atlas_size=2048, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    server = bt.AssetServer()
    paths = [f"face{s}" for s in range(6)]
    for p, f in zip(paths, faces):
        server.insert(p, f)

    def queue_jobs(pre, atlas, defer):
        pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, 5)), server, atlas, defer_upload=defer)

    return end_to_end_example(device, cfg, queue_jobs, passes)
