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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402,F401

import bevy_terrain_amd as bt  # noqa: E402


import workloads as W  # noqa: E402  (tools/workloads.py: the jobs themselves, shared with bench.py's `config.workloads`)

STEPS = None  # --steps N: fewer steps per job (counter passes under rocprofv3)
ONLY = None   # --only NAME: one workload (config2_height_4k / config2_albedo_4k / config5_cube_height_8k / config3_masked_16k / config3_masked_16k_fresh)


def launches_of(prof):
    return W.launches_of(prof)


def time_job(device, pre, atlas, steps=50):
    return W.time_job(device, pre, atlas, STEPS or steps, 2 if STEPS else 10)


def main():
    global STEPS, ONLY
    if "--steps" in sys.argv:
        STEPS = int(sys.argv[sys.argv.index("--steps") + 1])
    if "--only" in sys.argv:
        ONLY = sys.argv[sys.argv.index("--only") + 1]
    device = bt.Device(0)
    out = {}
    if "--end-to-end" in sys.argv:  # the reference's two examples, sources in host memory -> files written (preprocessor.rs:363,419), streamed and serial
        out["end_to_end_config2_planar_height_albedo"] = W.end_to_end_config2(device)
        out["end_to_end_config5_cube_height"] = W.end_to_end_config5(device)
        print(json.dumps(out))
        return
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
    if "--masked16k" in sys.argv or (ONLY or "").startswith("config3_masked_16k"):
        # --fresh (or --only config3_masked_16k_fresh): also each run on an atlas nothing has written since bt_atlas_create (6 of them)
        fresh = "--fresh" in sys.argv or ONLY == "config3_masked_16k_fresh"
        # (--clean: the same passes over the raster WITHOUT the mask — what a fresh, memset atlas alone does to the counters)
        print(json.dumps(W.masked16k(device, STEPS or 20, fresh_atlases=(max(4, STEPS or 6) if fresh else 0), rerun=ONLY != "config3_masked_16k_fresh", mask="--clean" not in sys.argv)))
        return
    if not ONLY or ONLY.startswith("config2"):  # (--only config5_...: no config 2 launch in the process — counter passes average per kernel name)
        out.update(W.config2(device, STEPS or 50, only=ONLY))
    if "--config2" in sys.argv or (ONLY and ONLY.startswith("config2")):
        print(json.dumps(out))
        return
    if "--cube-albedo" in sys.argv or ONLY == "config5_cube_albedo_8k":  # config 5's second attachment: 6 faces of 8192^2 Rgba8 (1.6 GB of source), lod_count 5
        print(json.dumps(W.config5_albedo(device, STEPS or 20)))
        return
    out.update(W.config5_height(device, STEPS or 50))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
