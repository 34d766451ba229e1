#!/usr/bin/env python3
"""Timing of the non-headline BASELINE configs (parity cases, not bench lines): config 2 (4k height + albedo)
and config 5 (cube, 6 x 8192^2 height).  Prints ms per job and the launch profile."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import bevy_terrain_amd as bt  # noqa: E402


STEPS = None  # --steps N: fewer steps per job (counter passes under rocprofv3)
ONLY = None   # --only NAME: one workload (config2_height_4k / config2_albedo_4k / config5_cube_height_8k / config3_masked_16k)


def launches_of(prof):
    return [(l["kind"], round(l["avg_ms"] * 1e3, 1), l["algorithmic_bytes"]) for l in prof]


def time_job(device, pre, atlas, steps=50):
    if STEPS:
        steps = STEPS
    for _ in range(10 if not STEPS else 2):
        pre.run(atlas, keep_queue=True, sync=False)
    device.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(device.torch_stream)
    for _ in range(steps):
        pre.run(atlas, keep_queue=True, sync=False, profile=True)
    e.record(device.torch_stream)
    device.synchronize()
    return s.elapsed_time(e) / steps, pre.profile(), pre.stats()


def masked_16k(device):
    """The 16k job with the 5 % no-data mask of tests/test_gpu_preprocess.py (seed 43: 37 x 53 texel cells + single texels): about
    half of all 8-row chunks hold a no-data texel and are redone by the generic rows."""
    size, lods = 16384, 6
    ptr = device.synth_fbm_r16(size, size, 42)
    src = device.download(ptr, (size, size), np.uint16)
    rng = np.random.default_rng(43)
    cells = rng.random((size // 37 + 1, size // 53 + 1)) < 0.05
    mask = np.repeat(np.repeat(cells, 37, axis=0), 53, axis=1)[:size, :size]
    single = rng.integers(0, size, size=(size, 2))
    mask[single[:, 0], single[:, 1]] = True
    src[mask] = 0
    device.free(ptr)
    ptr = device.upload(src)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/masked16k", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("m", (ptr, size, size))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="m", lod_range=range(0, lods)), server, atlas)
    ms, prof, st = time_job(device, pre, atlas, steps=20)
    return {"ms": ms, "tiles": st["tiles"], "launches": launches_of(prof)}


def main():
    global STEPS, ONLY
    if "--steps" in sys.argv:
        STEPS = int(sys.argv[sys.argv.index("--steps") + 1])
    if "--only" in sys.argv:
        ONLY = sys.argv[sys.argv.index("--only") + 1]
    device = bt.Device(0)
    out = {}
    if "--big32k" in sys.argv:  # one step past BASELINE's largest input: 32768^2 R16, lod_count 7, 5461 tiles (four generations of fused_main's workgroups)
        size, lods = 32768, 7
        ptr = device.synth_fbm_r16(size, size, 77)
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=5500, path="terrains/big", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
            bt.PreprocessDataset(attachment_index=0, path="b", lod_range=range(0, lods)), bt.AssetServer().insert("b", (ptr, size, size)), atlas)
        ms, prof, st = time_job(device, pre, atlas, steps=20)
        print(json.dumps({"big_32k": {"ms": ms, "tiles": st["tiles"], "algorithmic_bytes": st["algorithmic_bytes"], "GBps": st["algorithmic_bytes"] / ms / 1e6,
                                      "launches": launches_of(prof)}}))
        return
    if "--masked16k" in sys.argv or ONLY == "config3_masked_16k":
        print(json.dumps({"config3_masked_16k": masked_16k(device)}))
        return
    # config 2
    h = device.synth_fbm_r16(4096, 4096, 1234)
    rng = np.random.default_rng(1235)
    albedo = rng.integers(1, 256, size=(4096, 4096, 4), dtype=np.uint8)
    cfg = bt.TerrainConfig(lod_count=4, path="terrains/planar", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=512, border_size=2, format=bt.AttachmentFormat.Rgba8))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("h", (h, 4096, 4096)).insert("a", albedo)
    for name, att, path in (("config2_height_4k", 0, "h"), ("config2_albedo_4k", 1, "a")):
        if ONLY and ONLY != name:
            continue
        pre = bt.Preprocessor.new().clear_attachment(att, atlas).preprocess_tile(
            bt.PreprocessDataset(attachment_index=att, path=path, lod_range=range(0, 4)), server, atlas)
        ms, prof, st = time_job(device, pre, atlas)
        out[name] = {"ms": ms, "tiles": st["tiles"], "launches": launches_of(prof)}
    if "--config2" in sys.argv or (ONLY and ONLY.startswith("config2")):
        print(json.dumps(out))
        return
    if "--cube-albedo" in sys.argv:  # config 5's second attachment: 6 faces of 8192^2 Rgba8 (1.6 GB of source), lod_count 5
        rng = np.random.default_rng(77)
        face = rng.integers(1, 256, size=(8192, 8192, 4), dtype=np.uint8)
        cfg = bt.TerrainConfig(lod_count=5, atlas_size=2048, path="terrains/spherical")
        cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=512, border_size=2, format=bt.AttachmentFormat.Rgba8))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer()
        paths = [f"albedo{s}" for s in range(6)]
        for i, p in enumerate(paths):
            server.insert(p, np.roll(face, 997 * i, axis=1))
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
            bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, 5)), server, atlas)
        ms, prof, st = time_job(device, pre, atlas, steps=20)
        print(json.dumps({"config5_cube_albedo_8k": {"ms": ms, "tiles": st["tiles"], "launches": launches_of(prof)}}))
        return
    # config 5 (height)
    faces = [(device.synth_fbm_r16(8192, 8192, 7 + s), 8192, 8192) for s in range(6)]
    cfg = bt.TerrainConfig(lod_count=5, atlas_size=2048, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    paths = [f"face{s}" for s in range(6)]
    for p, f in zip(paths, faces):
        server.insert(p, f)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, 5)), server, atlas)
    ms, prof, st = time_job(device, pre, atlas)
    out["config5_cube_height_8k"] = {"ms": ms, "tiles": st["tiles"], "launches": launches_of(prof)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
