#!/usr/bin/env python3
"""How far do the workgroups of fused_main drift apart?  The profiling build (BT_FUSED_ABLATE = 134217728) stamps the shader
clock at chunks 0, 16, 32, 48 and at the end of every tile into the atlas's last layer; this tool runs the 16k job and prints,
per checkpoint, the spread of the stamps inside a tile row (32 workgroups that stream the same raster rows, one XCD) and
over the whole grid.  (The stamps of different XCDs are not guaranteed to share an origin: the in-row figures are the ones
to read.)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from bevy_terrain_amd import _ffi

_ffi.LIB_PATH = os.path.join(ROOT, "tools", "libbevy_terrain_amd_dbg.so")
os.environ["BT_FUSED_ABLATE"] = str(134217728 + int(os.environ.get("BT_PROBE_EXTRA", "0")))
import numpy as np

import bevy_terrain_amd as bt

SIZE, T, B, LODS, ATLAS = 16384, 512, 2, 6, 2048


def main():
    import torch

    torch.cuda.set_device(0)
    device = bt.Device(0)
    src = device.synth_fbm_r16(SIZE, SIZE, 42)
    cfg = bt.TerrainConfig(lod_count=LODS, atlas_size=ATLAS, path="terrains/drift", model=bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=B, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("src", (src, SIZE, SIZE))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas)
    pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, LODS)), server, atlas)
    for _ in range(30):
        pre.run(atlas, keep_queue=True, sync=False)
    device.synchronize()
    for rep in range(2):
        pre.run(atlas, keep_queue=True, sync=True)
        raw = atlas.download_tiles(0, ATLAS - 1, 1)[0]
        stamps = raw.reshape(-1).view(np.uint64)[: 1024 * 8].reshape(1024, 8)[:, :5].astype(np.int64)
        # items are in tile-row order: item = ty * 32 + tx
        t = stamps.reshape(32, 32, 5)
        clock_mhz = 100.0  # s_memrealtime: the constant 100 MHz reference clock, one origin for the whole device
        print(f"run {rep}: kernel span (first start -> last end, all XCDs) {(t[..., 4].max() - t[..., 0].min()) / clock_mhz:.1f} us")
        t0 = t[..., 0].min()
        dur = (t[..., 4] - t[..., 0]) / clock_mhz
        print(f"   first stamp (prologue + first staged rows done) after the earliest: median {np.median(t[..., 0] - t0) / clock_mhz:.1f} us, max {(t[..., 0].max() - t0) / clock_mhz:.1f} us;"
              f"   end stamp: min {(t[..., 4].min() - t0) / clock_mhz:.1f}, median {np.median(t[..., 4] - t0) / clock_mhz:.1f}, max {(t[..., 4].max() - t0) / clock_mhz:.1f} us")
        print(f"   per-tile duration (chunk 0 -> end): min {dur.min():.1f}, 10 % {np.percentile(dur, 10):.1f}, median {np.median(dur):.1f}, 90 % {np.percentile(dur, 90):.1f}, max {dur.max():.1f} us")
        per_xcd = [(float(np.median(t[4 * x:4 * x + 4, :, 0] - t0) / clock_mhz), float(np.median(t[4 * x:4 * x + 4, :, 4] - t0) / clock_mhz), float((t[4 * x:4 * x + 4, :, 4].max() - t0) / clock_mhz)) for x in range(8)]
        print("   per XCD (4 tile rows each): median start / median end / last end:", "  ".join(f"{a:.0f}/{b:.0f}/{c:.0f}" for a, b, c in per_xcd))
        per_row_end = (np.median(t[..., 4], axis=1) - t0) / clock_mhz
        print("   median end per tile row:", " ".join(f"{v:.0f}" for v in per_row_end))
        per_col_end = (np.median(t[..., 4], axis=0) - t0) / clock_mhz
        print("   median end per tile column:", " ".join(f"{v:.0f}" for v in per_col_end))
        for c, name in enumerate(("chunk 0", "chunk 16", "chunk 32", "chunk 48", "end")):
            in_row = (t[..., c].max(axis=1) - t[..., c].min(axis=1)) / clock_mhz
            print(f"   {name:9s} spread inside a tile row: median {np.median(in_row):6.2f} us, max {in_row.max():6.2f} us;"
                  f"   per-tile duration so far: median {np.median(t[..., c] - t[..., 0]) / clock_mhz:6.1f} us")


if __name__ == "__main__":
    main()
