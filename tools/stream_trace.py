#!/usr/bin/env python3
"""Timeline of bt_preprocessor_run_streamed (profiling build, `make -C bevy_terrain_amd/csrc debug`: BT_STREAM_TRACE=1 prints host stamps of
the launcher and the saver thread):  tools/stream_trace.py [16k | config2 | cube]"""
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bevy_terrain_amd import _ffi  # noqa: E402

_ffi.LIB_PATH = os.environ.get("BT_LIB") or os.path.join(ROOT, "tools", "libbevy_terrain_amd_dbg.so")
import numpy as np  # noqa: E402

import bevy_terrain_amd as bt  # noqa: E402
import workloads as W  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "16k"
device = bt.Device(0)
if which == "config2":
    h_dev, albedo = W.config2_sources(device)
    height = device.download(h_dev, (4096, 4096), np.uint16)
    cfg = W.planar_cfg(4, 1024, "t", [("height", bt.AttachmentFormat.R16), ("albedo", bt.AttachmentFormat.Rgba8)])
    server = bt.AssetServer().insert("h", height).insert("a", albedo)

    def queue(pre, atlas):
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, 4)), server, atlas, defer_upload=True)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=1, path="a", lod_range=range(0, 4)), server, atlas, defer_upload=True)
elif which == "cube":
    faces = [device.download(p, (h, w), np.uint16) for p, w, h in W.cube_faces(device)]
    cfg = bt.TerrainConfig(lod_count=5, atlas_size=2048, path="t")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    server = bt.AssetServer()
    paths = [f"f{s}" for s in range(6)]
    for p, f in zip(paths, faces):
        server.insert(p, f)

    def queue(pre, atlas):
        pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, 5)), server, atlas, defer_upload=True)
else:
    ptr = device.synth_fbm_r16(16384, 16384, 42)
    host = device.download(ptr, (16384, 16384), np.uint16)
    cfg = W.planar_cfg(6, 2048, "t", [("height", bt.AttachmentFormat.R16)])
    server = bt.AssetServer().insert("h", host)

    def queue(pre, atlas):
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, 6)), server, atlas, defer_upload=True)

for i in range(4):
    if i == 3:
        os.environ["BT_STREAM_TRACE"] = "1"
    root = tempfile.mkdtemp(prefix="bt_trace_", dir="/dev/shm")
    atlas = bt.TileAtlas.new(cfg, device)
    device.synchronize()
    pre = bt.Preprocessor.new()
    for ai in range(len(cfg.attachments)):
        pre.clear_attachment(ai, atlas, root)
    t0 = time.perf_counter()
    queue(pre, atlas)
    t1 = time.perf_counter()
    st = pre.run_streamed(atlas, root)
    print("pass", i, "queue", round((t1 - t0) * 1e3, 2), "total", round((time.perf_counter() - t0) * 1e3, 2), "ms", st, flush=True)
    pre.close()
    atlas.close()
    shutil.rmtree(root, ignore_errors=True)
