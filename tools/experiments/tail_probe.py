#!/usr/bin/env python3
"""Where does a workgroup of fused_tail (R16) spend its time on the 16k job?  The profiling build (BT_FUSED_ABLATE = 134217728) stamps the
100 MHz real-time clock per workgroup at entry, when lod-1 is computed (its loads have landed), when lod-1 is stored, when lod-2 is stored and
at the end; this tool runs the headline job and prints each phase over the 1024 mosaic workgroups."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bevy_terrain_amd import _ffi

_ffi.LIB_PATH = os.environ.get("BT_LIB") or os.path.join(ROOT, "tools", "libbevy_terrain_amd_dbg.so")
os.environ["BT_FUSED_ABLATE"] = str(134217728 + int(os.environ.get("BT_PROBE_EXTRA", "0")))
import numpy as np

import bevy_terrain_amd as bt


def main():
    import torch

    torch.cuda.set_device(0)
    device = bt.Device(0)
    size, lods = 16384, 6
    h = device.synth_fbm_r16(size, size, 42)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/probe", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("h", (h, size, size))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, lods)), server, atlas)
    for _ in range(20):
        pre.run(atlas, keep_queue=True, sync=False)
    device.synchronize()
    n = 32 * (16 if os.environ.get("BT_FUSED_TAIL_BLOCKS", "2") == "2" else 32)  # mosaic workgroups (32 x 16 with two blocks per thread); the apron workgroups behind them stamp nothing
    for rep in range(2):
        pre.run(atlas, keep_queue=True, sync=True)
        raw = atlas.download_tiles(0, 2047, 1)[0]
        t = raw.reshape(-1).view(np.uint64)[32768: 32768 + n * 8].reshape(n, 8)[:, :5].astype(np.int64) / 100.0  # us
        t0 = t[:, 0].min()
        q = lambda v: f"min {v.min():5.1f}  10 % {np.percentile(v, 10):5.1f}  median {np.median(v):5.1f}  90 % {np.percentile(v, 90):5.1f}  max {v.max():5.1f}"
        print(f"run {rep}: span first entry -> last end {t[:, 4].max() - t0:.1f} us over {n} mosaic workgroups")
        print("   entry after the first       :", q(t[:, 0] - t0))
        print("   entry -> lod-1 computed     :", q(t[:, 1] - t[:, 0]))
        print("   lod-1 stores (+ pushes)     :", q(t[:, 2] - t[:, 1]))
        print("   lod-2 computed + stored     :", q(t[:, 3] - t[:, 2]))
        print("   lod-3                       :", q(t[:, 4] - t[:, 3]))
        print("   end after the first entry   :", q(t[:, 4] - t0))


if __name__ == "__main__":
    main()
