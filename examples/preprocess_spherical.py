#!/usr/bin/env python3
"""examples/preprocess_spherical.rs on the MI355X library (the same configuration and builder calls): six cube-face height rasters -> 2046 tiles of 512^2
(LODs 0-4) with the cross-face borders stitched, `assets/terrains/spherical/data/height/*.bin` + `config.tc`.

    python examples/preprocess_spherical.py [--assets DIR] [--size 2048]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_terrain_amd as bt  # noqa: E402
from bevy_terrain_amd import AssetServer, AttachmentConfig, AttachmentFormat, Preprocessor, SphericalDataset, TerrainConfig, TileAtlas  # noqa: E402

import _sources  # noqa: E402

PATH = "terrains/spherical"
TEXTURE_SIZE = 512
LOD_COUNT = 5


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default="assets")
    ap.add_argument("--size", type=int, default=2048, help="side of the synthesised faces when the real ones are absent (GEBCO's are 8192+)")
    args = ap.parse_args()
    device = bt.Device(0)
    asset_server = AssetServer(args.assets)
    for side in range(6):
        _sources.ensure(os.path.join(args.assets, PATH, "source", "height", f"face{side}.tif"), lambda side=side: _sources.height(device, args.size, 7 + side))

    config = (TerrainConfig(lod_count=LOD_COUNT, path=PATH, atlas_size=2048)
              .add_attachment(AttachmentConfig(name="height", texture_size=TEXTURE_SIZE, border_size=2, format=AttachmentFormat.R16)))

    tile_atlas = TileAtlas.new(config, device)

    t0 = time.perf_counter()
    preprocessor = (Preprocessor.new()
                    .clear_attachment(0, tile_atlas, args.assets)
                    .preprocess_spherical(SphericalDataset(attachment_index=0,
                                                           paths=[f"{PATH}/source/height/face{side}.tif" for side in range(6)],
                                                           lod_range=range(0, LOD_COUNT)),
                                          asset_server, tile_atlas, defer_upload=True))
    t1 = time.perf_counter()
    stats = preprocessor.run_streamed(tile_atlas, args.assets)
    t2 = time.perf_counter()

    d = tile_atlas.attachment_directory(args.assets, 0)
    print(f"Preprocessing took {t2 - t1:.3f} seconds.  (sources decoded and queued in {t1 - t0:.3f} s)")
    print(f"  height: {len(os.listdir(d))} tiles in {d}; {stats['bands']} bands, {stats['uploaded_bytes'] >> 20} MiB up, {stats['saved_bytes'] >> 20} MiB down")


if __name__ == "__main__":
    main()
