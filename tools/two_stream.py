#!/usr/bin/env python3
"""Experiment: the 16k job as a pipeline of independent jobs on TWO contexts (two HIP streams, two atlases).
Does the 30 us fused_tail of one job hide behind the fused_main of the next?  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402,F401

import bevy_terrain_amd as bt  # noqa: E402


def job(device, src, size, lods):
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=1400, path="terrains/two", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("h", (src, size, size))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
        bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, lods)), server, atlas)
    return atlas, pre


def main():
    size, lods, steps = 16384, 6, 200
    devs = [bt.Device(0), bt.Device(0)]
    src = devs[0].synth_fbm_r16(size, size, 42)
    devs[0].synchronize()
    jobs = [job(d, src, size, lods) for d in devs]
    out = {}
    for mode, order in (("one_stream", [0] * steps), ("two_streams_alternating", [i & 1 for i in range(steps)])):
        for i in order[:20]:
            jobs[i][1].run(jobs[i][0], keep_queue=True, sync=False)
        for d in devs:
            d.synchronize()
        t0 = time.perf_counter()
        for i in order:
            jobs[i][1].run(jobs[i][0], keep_queue=True, sync=False)
        for d in devs:
            d.synchronize()
        out[mode] = {"ms_per_step": (time.perf_counter() - t0) * 1e3 / steps}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
