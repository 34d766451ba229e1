#!/usr/bin/env python3
"""Does fused_main's time depend on WHERE its buffers are?  One process, several (source, atlas) allocations kept alive side by
side (so every trial gets other addresses), the 16k job timed on each; prints the device addresses next to the times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: F401
import bevy_terrain_amd as bt


def job(device, src, atlas):
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
        bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, 6)), bt.AssetServer().insert("h", (src, 16384, 16384)), atlas)
    for _ in range(20):
        pre.run(atlas, keep_queue=True, sync=False)
    device.synchronize()
    for _ in range(60):
        pre.run(atlas, keep_queue=True, sync=False, profile=True)
    device.synchronize()
    prof = {l["kind"]: round(l["avg_ms"] * 1e3, 1) for l in pre.profile()}
    pre.close()
    return prof


def main():
    device = bt.Device(0)
    cfg = bt.TerrainConfig(lod_count=6, atlas_size=2048, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    keep = []
    pads = [0, 0, 3, 0, 517, 0, 33, 0]  # MB allocated (and kept) before the trial's buffers
    for trial, pad in enumerate(pads):
        if pad:
            keep.append(device.malloc(pad << 20))
        src = device.synth_fbm_r16(16384, 16384, 42)
        atlas = bt.TileAtlas.new(cfg, device)
        base = atlas.attachment_storage(0)[0]
        prof = job(device, src, atlas)
        again = job(device, src, atlas)
        print(f"trial {trial}: pad {pad:4d} MB  source {src:#014x}  atlas {base:#014x}  fused_main {prof['fused_main']:6.1f} / {again['fused_main']:6.1f} us  tail {prof['fused_tail']}", flush=True)
        keep.append((src, atlas))


main()
