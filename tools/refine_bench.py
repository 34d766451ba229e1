#!/usr/bin/env python3
"""Times the tiling prepass (one persistent launch per frame) on scripted camera paths; prints one JSON line.
Not the headline benchmark (bench.py) — refinement is latency-bound: 10^2..10^4 tiles x 16 B per frame."""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import bevy_terrain_amd as bt


def time_frames(device, prepass, views, **form):
    """device time per frame: the best of three repeats of each frame (single samples carry the odd 10-20 us of launch jitter)"""
    ms = []
    for v in views:
        best = None
        for _ in range(3):
            device.timer_begin()
            prepass.run(v, **form)
            t = device.timer_end()
            best = t if best is None or t < best else best
        ms.append(best)
    return ms


def measure(device, sweep=False):
    out = {}
    for name, model, positions in (
        ("planar_side1000", bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 250.0),
         [(700.0 * (1 - 0.9 * t) * math.cos(19 * t), 900.0 - 770.0 * t, 700.0 * (1 - 0.9 * t) * math.sin(19 * t)) for t in np.linspace(0, 1, 64)]),
        ("sphere_earth", bt.TerrainModel.sphere((0.0, 0.0, 0.0), 6371000.0, -12000.0, 9000.0),
         [tuple(np.array([0.3 + math.cos(9 * t) * (1 - t), 0.9, 0.2 + math.sin(9 * t) * (1 - t)]) /
                np.linalg.norm([0.3 + math.cos(9 * t) * (1 - t), 0.9, 0.2 + math.sin(9 * t) * (1 - t)]) * (6371000.0 + 4.0e6 * (1 - t) + 2.0e3))
          for t in np.linspace(0, 1, 64)]),
    ):
        cfg = bt.TerrainViewConfig()
        prepass = bt.TilingPrepass(device, cfg.geometry_tile_count)
        views = [bt.make_view_state(model, cfg, p) for p in positions]
        for v in views[:4]:
            prepass.run(v)
        device.synchronize()
        ms = time_frames(device, prepass, views)
        counts = []
        for v in views:
            prepass.run(v)
            counts.append(len(prepass.read()[0]))
        # the plain single-launch form (every divide test inside its pass): the checker, and what rounds 1-2 shipped
        for v in views[:4]:
            prepass.run(v, plain=True)
        device.synchronize()
        ms_p, same = time_frames(device, prepass, views, plain=True), True
        for v, n in zip(views, counts):
            prepass.run(v, plain=True)
            same = same and len(prepass.read()[0]) == n
        # the unordered form (the reference's contract is the set): two chip-wide launches, no chain of passes
        for v in views[:4]:
            prepass.run(v, unordered=True)
        device.synchronize()
        ms_u, same_u = time_frames(device, prepass, views, unordered=True), True
        for v, n in zip(views, counts):
            prepass.run(v, unordered=True)
            same_u = same_u and len(prepass.read()[0]) == n
        radius_sweep = {}
        for radius in ((8, 12, 16, 20, 24, 28) if sweep else ()):
            prepass.set_window(radius)
            for v in views[:4]:
                prepass.run(v, unordered=True)
            device.synchronize()
            t_r = time_frames(device, prepass, views, unordered=True)
            radius_sweep[radius] = [round(1e3 * float(np.mean(t_r)), 1), round(1e3 * float(np.max(t_r)), 1)]
        prepass.set_window(0)
        # the CPU side of the same frame in the reference (TileTree::update over sides x lods x tree_size^2 nodes, f64):
        # here one launch + the read-back of the request / release lists (host wall time per update, synchronous)
        lods = 12
        tcfg = bt.TerrainConfig(lod_count=lods, atlas_size=16, path="terrains/none", model=model)
        tcfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=16, border_size=2))
        atlas = bt.TileAtlas.new(tcfg, device)
        tree = bt.TileTree(atlas, model, lods, cfg)
        for p in positions[:4]:
            tree.update(p)
        tree_us, requests = [], []
        for p in positions:
            t0 = time.perf_counter()
            released, requested = tree.update(p)
            tree_us.append((time.perf_counter() - t0) * 1e6)
            requests.append(len(released) + len(requested))
        # the whole per-frame chain (plugin.rs:46-56): as separate calls (a synchronisation in update, the height read back,
        # the view state derived on the host, the prepass enqueued) and as ONE call (bt_frame_update: one synchronisation, the
        # height stays on the device); host wall time per frame, the stream drained at the end of every frame in both
        chain_us, frame_us = [], []
        fp = bt.TilingPrepass(device, cfg.geometry_tile_count)
        for p in positions[:4]:
            tree.frame_update(p, fp, unordered=True)
        device.synchronize()
        for p in positions:
            t0 = time.perf_counter()
            tree.update(p)
            tree.apply_requests()
            tree.adjust_to_tile_atlas()
            tree.approximate_height()
            fp.run(tree.view_state(), unordered=True)
            device.synchronize()
            chain_us.append((time.perf_counter() - t0) * 1e6)
        for p in positions:
            t0 = time.perf_counter()
            tree.frame_update(p, fp, unordered=True)
            device.synchronize()
            frame_us.append((time.perf_counter() - t0) * 1e6)
        tree.close()
        out[name + "_tile_tree"] = {"nodes": tree.nodes, "lod_count": lods, "us_per_update_avg_host_wall": float(np.mean(tree_us)),
                                     "us_per_update_max_host_wall": float(np.max(tree_us)), "requests_plus_releases_avg": float(np.mean(requests)),
                                     "frame_chain_separate_calls_us_avg_host_wall": float(np.mean(chain_us)), "frame_chain_separate_calls_us_max": float(np.max(chain_us)),
                                     "frame_update_one_call_us_avg_host_wall": float(np.mean(frame_us)), "frame_update_one_call_us_max": float(np.max(frame_us)),
                                     "frame_note": "update -> apply requests -> adjust_to_tile_atlas -> approximate_height -> unordered prepass, through the Python binding, stream drained per frame"}
        out[name] = {"frames": len(views), "us_per_frame_avg": 1e3 * float(np.mean(ms)), "us_per_frame_max": 1e3 * float(np.max(ms)),
                     "final_tiles_avg": float(np.mean(counts)), "final_tiles_max": int(np.max(counts)),
                     "launches_per_frame": 2, "reference_dispatches_per_frame": 2 * cfg.refinement_count + 3,
                     "unordered": {"us_per_frame_avg": 1e3 * float(np.mean(ms_u)), "us_per_frame_max": 1e3 * float(np.max(ms_u)), "launches_per_frame": 2,
                                   "same_tile_counts": bool(same_u), "window_radius_sweep_avg_max_us": radius_sweep,
                                   "note": "bt_tiling_prepass_run_unordered: the same set in arrival order (as the reference's atomics), every tile decided from its ancestors' divide bits"},
                     "plain_single_launch": {"us_per_frame_avg": 1e3 * float(np.mean(ms_p)), "us_per_frame_max": 1e3 * float(np.max(ms_p)), "launches_per_frame": 1,
                                             "same_tile_counts": bool(same),
                                             "note": "bt_tiling_prepass_run_plain: every divide test evaluated inside the pass that needs it"}}
    return out


def main():
    print(json.dumps({"tiling_prepass": measure(bt.Device(0), sweep="--sweep" in sys.argv)}))


if __name__ == "__main__":
    main()
