#!/usr/bin/env python3
"""examples/spherical.rs without the renderer: the cube-sphere terrain of preprocess_spherical.py on the WGS84 ellipsoid, one view
approaching from three radii out (the reference's DebugCameraBundle starts at -X * RADIUS * 3) down to a few kilometres above the surface —
per frame: TileTree::update over 6 sides x 16 LODs, tile requests / releases, streamed tile loads with their mip chains, adjust_to_tile_atlas,
approximate_height, and the tiling prepass that leaves the final tile list + indirect draw arguments in HBM.

    python examples/preprocess_spherical.py && python examples/spherical.py [--assets DIR] [--frames 120]
"""
import argparse
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bevy_terrain_amd as bt  # noqa: E402
from bevy_terrain_amd import (AttachmentConfig, AttachmentFormat, TerrainConfig, TerrainModel, TerrainViewConfig, TileAtlas, TileTree,  # noqa: E402
                              TilingPrepass)

PATH = "terrains/spherical"
RADIUS = 6371000.0
MAJOR_AXES = 6378137.0
MINOR_AXES = 6356752.314245
MIN_HEIGHT = -12000.0
MAX_HEIGHT = 9000.0
TEXTURE_SIZE = 512
LOD_COUNT = 16


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--assets", default="assets")
    ap.add_argument("--frames", type=int, default=120)
    args = ap.parse_args()
    if not os.path.exists(os.path.join(args.assets, PATH, "config.tc")):
        sys.exit(f"{args.assets}/{PATH}/config.tc not found: run examples/preprocess_spherical.py first")
    device = bt.Device(0)

    # Configure all the important properties of the terrain, as well as its attachments.
    config = (TerrainConfig(lod_count=LOD_COUNT, model=TerrainModel.ellipsoid((0.0, 0.0, 0.0), MAJOR_AXES, MINOR_AXES, MIN_HEIGHT, MAX_HEIGHT), path=PATH)
              .add_attachment(AttachmentConfig(name="height", texture_size=TEXTURE_SIZE, border_size=2, mip_level_count=4, format=AttachmentFormat.R16)))

    # Configure the quality settings of the terrain view. Adapt the settings to your liking.
    view_config = TerrainViewConfig()

    tile_atlas = TileAtlas.new(config, device)
    tile_atlas.load_tile_config(args.assets)
    tile_tree = TileTree.new(tile_atlas, view_config)
    prepass = TilingPrepass(device, view_config.geometry_tile_count)

    loaded_total, failed_total, requested_total, t_frames = 0, 0, 0, 0.0
    for frame in range(args.frames):
        # the debug camera: from -X * RADIUS * 3 down to 5 km above the surface, a quarter turn around the polar axis on the way
        t = frame / max(args.frames - 1, 1)
        distance = RADIUS * (1.0 + 2.0 * (1.0 - t) ** 3) + 5000.0
        phi = math.pi + 0.5 * math.pi * t
        view_position = (distance * math.cos(phi), 0.2 * distance * t, distance * math.sin(phi))
        t0 = time.perf_counter()
        loaded, failed = tile_atlas.update(args.assets)
        info = tile_tree.frame_update(view_position, prepass)
        t_frames += time.perf_counter() - t0
        loaded_total += loaded
        failed_total += failed
        requested_total += info.requested_count
        if frame % max(args.frames // 8, 1) == 0 or frame == args.frames - 1:
            tiles, indirect = prepass.read()
            lods = sorted({int(l) for l in tiles[:, 1]}) if len(tiles) else []
            print(f"frame {frame:4d}: altitude {(distance - RADIUS) / 1000.0:9.1f} km  requested {info.requested_count:3d} released {info.released_count:3d} loaded {loaded:3d}  "
                  f"final tiles {len(tiles):5d} on LODs {lods[0] if lods else '-'}..{lods[-1] if lods else '-'} (draw: {indirect[0]} vertices)  approximate height {info.approximate_height:8.1f}")
    print(f"{args.frames} frames, {requested_total} tile requests, {loaded_total} tile loads ({failed_total} failed), pending {tile_atlas.pending_loads()}, "
          f"{1e3 * t_frames / args.frames:.3f} ms of host time per frame (loads included)")


if __name__ == "__main__":
    main()
