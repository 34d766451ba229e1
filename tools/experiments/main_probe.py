#!/usr/bin/env python3
"""Where does a workgroup of fused_main spend its time on a SMALL job?  The profiling build (BT_FUSED_ABLATE = 134217728) stamps the
100 MHz real-time clock per workgroup at entry, after the prologue (tables, lookups, first staged rows landed), after the chunk loop and
at the end; this tool runs config 2's height job (4096^2 R16, 85 tiles: 64 finest tiles x 16 parts of 4 chunks) and prints each phase."""
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
    h = device.synth_fbm_r16(4096, 4096, 1234)
    cfg = bt.TerrainConfig(lod_count=4, atlas_size=128, path="terrains/probe", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("h", (h, 4096, 4096))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, 4)), server, atlas)
    for _ in range(30):
        pre.run(atlas, keep_queue=True, sync=False)
    device.synchronize()
    n = int(os.environ.get("BT_PROBE_WGS", "1024"))
    for rep in range(2):
        pre.run(atlas, keep_queue=True, sync=True)
        raw = atlas.download_tiles(0, 127, 1)[0]
        t = raw.reshape(-1).view(np.uint64)[16384: 16384 + n * 4].reshape(n, 4).astype(np.int64) / 100.0  # us
        t0 = t[:, 0].min()
        q = lambda v: f"min {v.min():5.1f}  10 % {np.percentile(v, 10):5.1f}  median {np.median(v):5.1f}  90 % {np.percentile(v, 90):5.1f}  max {v.max():5.1f}"
        print(f"run {rep}: span first entry -> last end {t[:, 3].max() - t0:.1f} us over {n} workgroups")
        print("   entry after the first    :", q(t[:, 0] - t0))
        print("   prologue                 :", q(t[:, 1] - t[:, 0]))
        print("   chunk loop               :", q(t[:, 2] - t[:, 1]))
        print("   epilogue (redo check)    :", q(t[:, 3] - t[:, 2]))
        print("   end after the first entry:", q(t[:, 3] - t0))


if __name__ == "__main__":
    main()
