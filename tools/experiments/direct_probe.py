#!/usr/bin/env python3
"""Where does a workgroup of fused_direct spend its time?  The profiling build (BT_FUSED_ABLATE = 134217728) stamps the 100 MHz
real-time clock per workgroup at entry, after the set-up, after each of the two sweeps and at the end, into the atlas's last layer;
this tool runs config 2's albedo job (4096^2 Rgba8, 85 tiles) and prints the distribution of each phase."""
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
    albedo = np.random.default_rng(1235).integers(1, 256, size=(4096, 4096, 4), dtype=np.uint8)
    cfg = bt.TerrainConfig(lod_count=4, atlas_size=128, path="terrains/probe", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=512, border_size=2, format=bt.AttachmentFormat.Rgba8))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("a", albedo)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="a", lod_range=range(0, 4)), server, atlas)
    for _ in range(30):
        pre.run(atlas, keep_queue=True, sync=False)
    device.synchronize()
    for rep in range(2):
        pre.run(atlas, keep_queue=True, sync=True)
        raw = atlas.download_tiles(0, 127, 1)[0]
        n = int(os.environ.get("BT_PROBE_WGS", "1024"))
        t = raw.reshape(-1).view(np.uint64)[: n * 8].reshape(n, 8)[:, :5].astype(np.int64) / 100.0  # us
        t0 = t[:, 0].min()
        q = lambda v: f"min {v.min():5.1f}  10 % {np.percentile(v, 10):5.1f}  median {np.median(v):5.1f}  90 % {np.percentile(v, 90):5.1f}  max {v.max():5.1f}"
        print(f"run {rep}: span first entry -> last end {t[:, 4].max() - t0:.1f} us over {n} workgroups")
        print("   entry after the first   :", q(t[:, 0] - t0))
        print("   set-up                  :", q(t[:, 1] - t[:, 0]))
        print("   sweep 0 (256 columns)   :", q(t[:, 2] - t[:, 1]))
        print("   sweep 1 (252 + aprons)  :", q(t[:, 3] - t[:, 2]))
        print("   apron rows              :", q(t[:, 4] - t[:, 3]))
        print("   end after the first entry:", q(t[:, 4] - t0))
        slow = np.argsort(t[:, 4])[-8:]
        print("   the last eight to finish (workgroup: entry / set-up / sweep 0 / sweep 1 / aprons / end):")
        for w in slow:
            print(f"      {w:5d}: {t[w, 0] - t0:5.1f} / {t[w, 1] - t[w, 0]:5.1f} / {t[w, 2] - t[w, 1]:5.1f} / {t[w, 3] - t[w, 2]:5.1f} / {t[w, 4] - t[w, 3]:5.1f} / {t[w, 4] - t0:5.1f}")


if __name__ == "__main__":
    main()
