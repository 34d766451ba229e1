#!/usr/bin/env python3
"""The register-staged variant of fused_main (rasters whose pitch is not a multiple of 16 bytes): a 16380^2 R16 job, clean and with the 5 % no-data
mask of config_bench.py, ms per job.  BT_LIB=<library> for A/B runs inside one lease."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from bevy_terrain_amd import _ffi

if os.environ.get("BT_LIB"):
    _ffi.LIB_PATH = os.environ["BT_LIB"]
import numpy as np
import torch  # noqa: F401

import bevy_terrain_amd as bt
from config_bench import time_job


def main():
    device = bt.Device(0)
    size, lods = 16380, 6
    ptr = device.synth_fbm_r16(size, size, 42)
    src = device.download(ptr, (size, size), np.uint16)
    device.free(ptr)
    out = {}
    for name in ("clean", "masked"):
        if name == "masked":
            rng = np.random.default_rng(43)
            cells = rng.random((size // 37 + 1, size // 53 + 1)) < 0.05
            src[np.repeat(np.repeat(cells, 37, axis=0), 53, axis=1)[:size, :size]] = 0
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/u", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
            bt.PreprocessDataset(attachment_index=0, path="m", lod_range=range(0, lods)), bt.AssetServer().insert("m", src), atlas)
        ms, prof, st = time_job(device, pre, atlas, steps=20)
        out[name] = {"ms": round(ms, 4), "launches": [(l["kind"], round(l["avg_ms"] * 1e3, 1)) for l in prof]}
        pre.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
