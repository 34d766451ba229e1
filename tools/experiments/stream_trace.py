#!/usr/bin/env python3
"""Timeline of bt_preprocessor_run_streamed on the 16k job (profiling build: BT_STREAM_TRACE=1 prints host stamps)."""
import os, sys, time, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bevy_terrain_amd import _ffi
_ffi.LIB_PATH = os.path.join(ROOT, "tools", "libbevy_terrain_amd_dbg.so")
import numpy as np
import bevy_terrain_amd as bt

device = bt.Device(0)
ptr = device.synth_fbm_r16(16384, 16384, 42)
host = device.download(ptr, (16384, 16384), np.uint16)
cfg = bt.TerrainConfig(lod_count=6, atlas_size=2048, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
atlas = bt.TileAtlas.new(cfg, device)
root = tempfile.mkdtemp(prefix="bt_trace_", dir="/dev/shm")
for i in range(3):
    if i == 2:
        os.environ["BT_STREAM_TRACE"] = "1"
    pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
    t0 = time.perf_counter()
    pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, 6)), bt.AssetServer().insert("h", host), atlas, defer_upload=True)
    st = pre.run_streamed(atlas, root)
    print("pass", i, round((time.perf_counter() - t0) * 1e3, 2), "ms", st, flush=True)
    pre.close()
shutil.rmtree(root, ignore_errors=True)
