#!/usr/bin/env python3
"""examples/minimal.rs without the renderer: the planar terrain of preprocess_planar.py, one view, and the per-frame work the
reference's plugins schedule for it (plugin.rs:46-56, tiling_prepass.rs:204-272) — TileTree::update -> tile requests / releases -> the
atlas streams the requested tiles from `data/height/*.bin` (mip chains built on the GPU) -> adjust_to_tile_atlas -> approximate_height ->
the tiling prepass (refine_tiles) that leaves the final tile list and the indirect draw arguments in HBM for whatever draws the terrain.
The debug camera of the reference is replaced by a scripted fly-over.

    python examples/preprocess_planar.py && python examples/minimal.py [--assets DIR] [--frames 120]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_terrain_amd as bt  # noqa: E402
from bevy_terrain_amd import (AttachmentConfig, AttachmentFormat, TerrainConfig, TerrainModel, TerrainViewConfig, TileAtlas, TileTree,  # noqa: E402
                              TilingPrepass, sample_height)

PATH = "terrains/planar"
TERRAIN_SIZE = 1000.0
HEIGHT = 250.0
TEXTURE_SIZE = 512
LOD_COUNT = 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default="assets")
    ap.add_argument("--frames", type=int, default=120)
    args = ap.parse_args()
    if not os.path.exists(os.path.join(args.assets, PATH, "config.tc")):
        sys.exit(f"{args.assets}/{PATH}/config.tc not found: run examples/preprocess_planar.py first")
    device = bt.Device(0)

    # Configure all the important properties of the terrain, as well as its attachments.
    config = (TerrainConfig(lod_count=LOD_COUNT, model=TerrainModel.planar((0.0, -100.0, 0.0), TERRAIN_SIZE, 0.0, HEIGHT), path=PATH)
              .add_attachment(AttachmentConfig(name="height", texture_size=TEXTURE_SIZE, border_size=2, mip_level_count=4, format=AttachmentFormat.R16)))

    # Configure the quality settings of the terrain view. Adapt the settings to your liking.
    view_config = TerrainViewConfig()

    tile_atlas = TileAtlas.new(config, device)
    tile_atlas.load_tile_config(args.assets)  # TileAtlas::new reads config.tc (tile_atlas.rs:311, 612-624)
    tile_tree = TileTree.new(tile_atlas, view_config)
    prepass = TilingPrepass(device, view_config.geometry_tile_count)  # TerrainViewData::new: final + temporary tile lists

    loaded_total, failed_total, requested_total, t_frames = 0, 0, 0, 0.0
    for frame in range(args.frames):
        # the debug camera: a descending spiral over the terrain
        t = frame / max(args.frames - 1, 1)
        r, phi = 0.45 * TERRAIN_SIZE * (1.0 - 0.8 * t), 4.0 * math.pi * t
        view_position = (r * math.cos(phi), -100.0 + HEIGHT + 600.0 * (1.0 - t) + 5.0, r * math.sin(phi))
        t0 = time.perf_counter()
        loaded, failed = tile_atlas.update(args.assets)                # finish the loads requested by earlier frames
        info = tile_tree.frame_update(view_position, prepass)         # update, requests, adjust, approximate height, tiling prepass
        t_frames += time.perf_counter() - t0
        loaded_total += loaded
        failed_total += failed
        requested_total += info.requested_count
        if frame % max(args.frames // 8, 1) == 0 or frame == args.frames - 1:
            tiles, indirect = prepass.read()
            ground = sample_height(tile_tree, tile_atlas, (view_position[0], 0.0, view_position[2]))
            print(f"frame {frame:4d}: view y {view_position[1]:7.1f}  requested {info.requested_count:3d} released {info.released_count:3d} "
                  f"loaded {loaded:3d}  final tiles {len(tiles):5d} (draw: {indirect[0]} vertices x {indirect[1]} instances)  height below the view {ground:6.1f}")
    print(f"{args.frames} frames, {requested_total} tile requests, {loaded_total} tile loads ({failed_total} failed), pending {tile_atlas.pending_loads()}, "
          f"{1e3 * t_frames / args.frames:.3f} ms of host time per frame (loads included)")


if __name__ == "__main__":
    main()
