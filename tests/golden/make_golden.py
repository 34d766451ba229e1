#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz FROM THE REFERENCE'S OWN SHADER TEXT.

The reference ships no golden vectors and its Rust / wgpu host cannot run here.  Its arithmetic,
however, is WGSL text under /root/reference/src/shaders, and oracle/wgsl_ref executes that text on the CPU
(wgsl2cpp.py: mechanical WGSL -> C++; ref_harness.cpp: bindings, texture unit, dispatch loop).  Every tile and every
tile list in these fixtures is an OUTPUT OF THAT EXECUTION:

  * split / downsample / stitch tiles: the oracle's queue driver (which restates the Rust side: task order, atlas index
    allocation, neighbour / child lists) hands each task to the executed split.wgsl / downsample.wgsl / stitch.wgsl;
  * tiling prepass lists: prepare_prepass.wgsl + refine_tiles.wgsl + functions.wgsl executed for scripted cameras; the
    view uniforms come from the oracle's restatement of the Rust f64 side (terrain_view_bind_group.rs, terrain_model.rs).

`formats.npz` (config.tc bytes, CPU mip chain) restates Rust code with no WGSL behind it and stays oracle-made.

The fixtures travel to the GPU box (which has no /root/reference); there tests/test_golden.py checks the hand-written
oracle AND the HIP product against them.

  python tests/golden/make_golden.py        (this container only: needs /root/reference)
"""
import hashlib
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402
import _wgslref as W  # noqa: E402

# name: (format, T, b, lod_count, H, W, seed, holes, datasets)
#   datasets: list of (top_left, bottom_right, lod_range, seed offset) — more than one = overlays onto the same attachment
FULL = [((0.0, 0.0), (1.0, 1.0), None, 0)]
CASES = {
    "planar_r16_t16": (O.FORMAT_R16, 16, 2, 3, 56, 53, 7001, 0.03, FULL),
    "planar_r16_t64": (O.FORMAT_R16, 64, 2, 3, 257, 300, 7002, 0.01, FULL),
    "planar_rgba8_t16": (O.FORMAT_RGBA8, 16, 2, 3, 53, 56, 7003, 0.03, FULL),
    "planar_r16_t128_one_hole": (O.FORMAT_R16, 128, 2, 3, 600, 640, 7004, -1.0, FULL),
    # a dataset that covers part of the terrain, and a second one laid over it (split.wgsl:37-42 keeps the previous
    # texel where the new dataset has no data or does not reach)
    "planar_r16_t32_subrect": (O.FORMAT_R16, 32, 4, 3, 90, 110, 7005, 0.05, [((0.125, 0.25), (0.8125, 0.9375), None, 0)]),
    "planar_r16_t24_overlay": (O.FORMAT_R16, 24, 2, 3, 70, 64, 7006, 0.2, [((0.0, 0.0), (1.0, 1.0), None, 0), ((0.25, 0.125), (0.75, 0.625), None, 1)]),
    "planar_rgba8_t32_b3": (O.FORMAT_RGBA8, 32, 3, 2, 61, 47, 7007, 0.1, FULL),
}
CUBE = {"cube_r16_t16": (O.FORMAT_R16, 16, 2, 2, 40, 7010, 0.02), "cube_rgba8_t16": (O.FORMAT_RGBA8, 16, 2, 2, 36, 7011, 0.05)}


def raster(fmt, h, w, seed, holes):
    rng = np.random.default_rng(seed)
    if fmt == O.FORMAT_R16:
        src = rng.integers(1, 65536, size=(h, w), dtype=np.uint16)
        if holes > 0:
            src[rng.random(src.shape) < holes] = 0
        elif holes < 0:
            src[h // 2 + 1, w // 2 + 3] = 0  # one isolated no-data texel
    else:
        src = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        src[..., 0] = np.maximum(src[..., 0], 1)
        if holes > 0:
            src[..., 0][rng.random(src.shape[:2]) < holes] = 0
    return src


def planar_job(atlas, fmt, lods, h, w, seed, holes, datasets):
    """queue the datasets of a case on `atlas`; returns the source rasters in dataset order"""
    sources = []
    atlas.clear_attachment(0)
    for tl, br, lod_range, ds in datasets:
        src = raster(fmt, h, w, seed + 100 * ds, holes)
        sources.append(src)
        atlas.preprocess_tile(0, src, lod_range or (0, lods), top_left=tl, bottom_right=br)
    return sources


def pack(atlas, extra, full=True):
    """coords: side, lod, x, y, atlas_index per tile; tile_sha256: one digest per tile; tiles: the texels (small cases)."""
    tiles = atlas.tiles()
    coords = np.array([list(c) + [i] for c, i in tiles], dtype=np.uint32)
    data = [np.ascontiguousarray(atlas.tile(0, i)) for _, i in tiles]
    digests = np.stack([np.frombuffer(hashlib.sha256(d.tobytes()).digest(), dtype=np.uint8) for d in data])
    out = dict(coords=coords, tile_sha256=digests, generator=np.array("oracle/_ref: executed WGSL"), wgsl_sha256=np.array(W.sources()), **extra)
    if full:
        out["tiles"] = np.stack(data)
    return out


def spiral(n, radius, h0, h1, seed=99):
    rng = np.random.default_rng(seed)
    for i in range(n):
        t = i / max(n - 1, 1)
        a = 2 * math.pi * 3 * t + rng.random() * 0.01
        r = radius * (1 - 0.9 * t)
        yield (r * math.cos(a), h0 + (h1 - h0) * t, r * math.sin(a))


def camera_paths():
    """(name, model args, approximate_height, positions) — the scripted cameras of SURVEY.md §8(d), shortened"""
    planar = [tuple(p) for p in spiral(10, 700.0, 900.0, 130.0)]
    sphere = []
    for x, h, z in spiral(8, 1.0, 4.0e6, 2.0e3):
        d = np.array([0.3 + x, 0.9, 0.2 + z])
        d = d / np.linalg.norm(d)
        sphere.append(tuple(d * (6371000.0 + h)))
    return [
        ("planar", dict(kind="planar", position=(0, 0, 0), a=1000.0, min_height=0.0, max_height=250.0), 100.0, planar),
        ("sphere", dict(kind="spherical", position=(0, 0, 0), a=6371000.0, min_height=-12000.0, max_height=9000.0), 500.0, sphere),
        ("ellipsoid", dict(kind="ellipsoidal", position=(0, 0, 0), a=6378137.0, b=6356752.314245, min_height=-12000.0, max_height=9000.0), 1500.0, sphere),
    ]


def main():
    if not os.path.isdir(W.REFERENCE):
        sys.exit("make_golden.py needs /root/reference (the fixtures are outputs of its WGSL)")
    W.build(force=True)
    for name, (fmt, T, b, lods, h, w, seed, holes, datasets) in CASES.items():
        a = W.attach(O.OracleAtlas(lods, 128, False, [(T, b, 1, fmt)]))
        sources = planar_job(a, fmt, lods, h, w, seed, holes, datasets)
        a.run(4)
        rects = np.array([[tl[0], tl[1], br[0], br[1]] for tl, br, _, _ in datasets], dtype=np.float32)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(a, dict(source=np.stack(sources), rects=rects, params=np.array([fmt, T, b, lods], dtype=np.uint32)), full=T <= 64))
    for name, (fmt, T, b, lods, w, seed, holes) in CUBE.items():
        faces = [raster(fmt, w, w, seed + s, holes) for s in range(6)]
        a = W.attach(O.OracleAtlas(lods, 128, True, [(T, b, 1, fmt)]))
        a.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(4)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(a, dict(source=np.stack(faces), params=np.array([fmt, T, b, lods], dtype=np.uint32))))
    # tiling prepass: final tile lists (append order of a sequential run), indirect args and per-pass counts
    vc = O.make_view_config(geometry_tile_count=100000)
    out = {}
    for name, margs, height, positions in camera_paths():
        model = O.make_model(**margs)
        views, lists, offsets, indirect, passes = [], [], [0], [], []
        for pos in positions:
            v = O.view_state_from_config(model, vc, pos, height)
            tiles, ind, pc = W.refine(v)
            views.append(np.frombuffer(bytes(v), dtype=np.uint8))
            lists.append(np.array(tiles, dtype=np.uint32).reshape(-1, 4))
            offsets.append(offsets[-1] + len(tiles))
            indirect.append(ind)
            passes.append(pc)
        out[name + "_views"] = np.stack(views)  # orc_view structs, byte for byte (what the uniforms were filled from)
        out[name + "_positions"] = np.array(positions, dtype=np.float64)
        out[name + "_tiles"] = np.concatenate(lists)
        out[name + "_offsets"] = np.array(offsets, dtype=np.int64)
        out[name + "_indirect"] = np.array(indirect, dtype=np.uint32)
        out[name + "_passes"] = np.array(passes, dtype=np.uint32)
    np.savez_compressed(os.path.join(HERE, "refine.npz"), generator=np.array("oracle/_ref: executed WGSL"), wgsl_sha256=np.array(W.sources()), **out)
    # config.tc bytes of a small coordinate set + a mip chain (Rust side only: oracle-made)
    coords = [(0, 0, 0, 0), (0, 1, 1, 0), (3, 5, 17, 250), (5, 12, 4095, 300), (1, 20, 70000, 5)]
    mip_src = raster(O.FORMAT_R16, 32, 32, 7020, 0.2)
    mips = O.generate_mipmaps(O.FORMAT_R16, mip_src, 4)
    np.savez_compressed(os.path.join(HERE, "formats.npz"), tc_coords=np.array(coords, dtype=np.uint32),
                        tc_bytes=np.frombuffer(O.tc_encode(coords), dtype=np.uint8), mip_source=mip_src,
                        mip_chain=np.asarray(mips))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
