#!/usr/bin/env python3
"""examples/preprocess_planar.rs on the MI355X library: the same configuration and the same builder calls.

The reference builds the queue in `setup` and lets Bevy's schedule drain it over many frames, saving tiles as they finish
(preprocessor.rs:345-422).  Here the same builder calls queue the same tasks and `run_streamed` is that whole span — source files decoded and
uploaded, every tile of both attachments produced, `assets/terrains/planar/data/{height,albedo}/*.bin` + `config.tc` written — as one
overlapped pipeline on the GPU.

    python examples/preprocess_planar.py [--assets DIR] [--size 4096]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_terrain_amd as bt  # noqa: E402
from bevy_terrain_amd import (AssetServer, AttachmentConfig, AttachmentFormat, PreprocessDataset, Preprocessor, TerrainConfig,  # noqa: E402
                              TerrainModel, TileAtlas)

import _sources  # noqa: E402

PATH = "terrains/planar"
TEXTURE_SIZE = 512
LOD_COUNT = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default="assets", help="Bevy's asset root")
    ap.add_argument("--size", type=int, default=4096, help="side of the synthesised sources when the real ones are absent")
    args = ap.parse_args()
    device = bt.Device(0)
    asset_server = AssetServer(args.assets)
    src = os.path.join(args.assets, PATH, "source")
    _sources.ensure(os.path.join(src, "height.png"), lambda: _sources.height(device, args.size, 1234))
    _sources.ensure(os.path.join(src, "albedo.png"), lambda: _sources.albedo_of(_sources.height(device, args.size, 1234)))

    # fn setup(...)
    config = (TerrainConfig(lod_count=LOD_COUNT, path=PATH, model=TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 250.0))
              .add_attachment(AttachmentConfig(name="height", texture_size=TEXTURE_SIZE, border_size=2, format=AttachmentFormat.R16))
              .add_attachment(AttachmentConfig(name="albedo", texture_size=TEXTURE_SIZE, border_size=2, format=AttachmentFormat.Rgba8)))

    tile_atlas = TileAtlas.new(config, device)

    t0 = time.perf_counter()
    preprocessor = (Preprocessor.new()
                    .clear_attachment(0, tile_atlas, args.assets)
                    .clear_attachment(1, tile_atlas, args.assets)
                    .preprocess_tile(PreprocessDataset(attachment_index=0, path=f"{PATH}/source/height.png", lod_range=range(0, LOD_COUNT)),
                                     asset_server, tile_atlas, defer_upload=True)
                    .preprocess_tile(PreprocessDataset(attachment_index=1, path=f"{PATH}/source/albedo.png", lod_range=range(0, LOD_COUNT)),
                                     asset_server, tile_atlas, defer_upload=True))
    t1 = time.perf_counter()
    # commands.spawn((tile_atlas, preprocessor)) -> the schedule runs the queue and saves the tiles
    stats = preprocessor.run_streamed(tile_atlas, args.assets)
    t2 = time.perf_counter()

    print(f"Preprocessing took {t2 - t1:.3f} seconds.  (sources decoded and queued in {t1 - t0:.3f} s)")  # preprocessor.rs:419
    for i, a in enumerate(config.attachments):
        d = tile_atlas.attachment_directory(args.assets, i)
        print(f"  {a.name}: {len(os.listdir(d))} tiles in {d}")
    print(f"  {stats['bands']} bands, {stats['uploaded_bytes'] >> 20} MiB up, {stats['saved_bytes'] >> 20} MiB down")


if __name__ == "__main__":
    main()
