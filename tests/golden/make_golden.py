#!/usr/bin/env python3
"""Regenerate tests/golden/*.npz.

The reference ships no golden vectors and cannot run here, so these fixtures are produced by
THIS repository's oracle (oracle/bt_oracle.c) on small seeded inputs.  They pin the oracle and the HIP product
against drift — a change in either that alters one output byte fails tests/test_golden.py — and let the GPU
tests compare against committed bytes without the oracle in the loop.  They do NOT pin the oracle to the
reference: "parity unpinned" stands.

  python tests/golden/make_golden.py        (needs only the CPU oracle)
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

CASES = {
    # name: (format, T, b, lod_count, H, W, seed, holes)
    "planar_r16_t16": (O.FORMAT_R16, 16, 2, 3, 56, 53, 7001, 0.03),
    "planar_r16_t64": (O.FORMAT_R16, 64, 2, 3, 257, 300, 7002, 0.01),
    "planar_rgba8_t16": (O.FORMAT_RGBA8, 16, 2, 3, 53, 56, 7003, 0.03),
    "planar_r16_t128_one_hole": (O.FORMAT_R16, 128, 2, 3, 600, 640, 7004, -1.0),
}
CUBE = {"cube_r16_t16": (O.FORMAT_R16, 16, 2, 2, 40, 7010, 0.02)}


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


def pack(atlas, extra, full=True):
    """coords: side, lod, x, y, atlas_index per tile; tile_sha256: one digest per tile; tiles: the texels (small cases)."""
    tiles = atlas.tiles()
    coords = np.array([list(c) + [i] for c, i in tiles], dtype=np.uint32)
    data = [np.ascontiguousarray(atlas.tile(0, i)) for _, i in tiles]
    digests = np.stack([np.frombuffer(hashlib.sha256(d.tobytes()).digest(), dtype=np.uint8) for d in data])
    out = dict(coords=coords, tile_sha256=digests, **extra)
    if full:
        out["tiles"] = np.stack(data)
    return out


def main():
    for name, (fmt, T, b, lods, h, w, seed, holes) in CASES.items():
        src = raster(fmt, h, w, seed, holes)
        a = O.OracleAtlas(lods, 128, False, [(T, b, 1, fmt)])
        a.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(4)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(a, dict(source=src, params=np.array([fmt, T, b, lods], dtype=np.uint32)), full=T <= 64))
    for name, (fmt, T, b, lods, w, seed, holes) in CUBE.items():
        faces = [raster(fmt, w, w, seed + s, holes) for s in range(6)]
        a = O.OracleAtlas(lods, 128, True, [(T, b, 1, fmt)])
        a.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(4)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **pack(a, dict(source=np.stack(faces), params=np.array([fmt, T, b, lods], dtype=np.uint32))))
    # config.tc bytes of a small coordinate set + a mip chain
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
