"""Cube-sphere adjacency: TileCoordinate::neighbours (coordinate.rs:208-279) and the WGSL
project_to_side tables (stitch.wgsl:12-51) restated in the oracle must describe the same cube."""
import numpy as np

import _oracle as O


def cube_position(side, u, v):
    """compute_local_position (functions.wgsl:73-96) without the sigmoid warp / normalisation."""
    u = 2.0 * u - 1.0
    v = 2.0 * v - 1.0
    return {
        0: (-1.0, -v, u), 1: (u, -v, 1.0), 2: (u, 1.0, v), 3: (1.0, -u, v), 4: (v, -u, -1.0), 5: (v, -1.0, u),
    }[side]


def face_raster(side, W, coeff=(8000.0, 5000.0, 3000.0)):
    t = (np.arange(W) + 0.5) / W
    out = np.zeros((W, W), np.float64)
    for ty in range(W):
        for tx in range(W):
            p = cube_position(side, t[tx], t[ty])
            out[ty, tx] = 30000.0 + coeff[0] * p[0] + coeff[1] * p[1] + coeff[2] * p[2]
    return np.round(out).astype(np.uint16)


def test_neighbours_are_symmetric_and_corners_invalid():
    for lod in (0, 1, 2, 3):
        n = 1 << lod
        for side in range(6):
            for x in range(n):
                for y in range(n):
                    c = (side, lod, x, y)
                    nb = O.neighbours(c, True)
                    assert len(nb) == 8
                    for k, m in enumerate(nb):
                        if m == (O.INVALID,) * 4:
                            # only diagonal neighbours at cube corners are missing
                            assert k >= 4 and x in (0, n - 1) and y in (0, n - 1)
                            continue
                        assert m[1] == lod and m[0] < 6 and m[2] < n and m[3] < n
                        if k < 4:  # edge neighbours are mutual
                            assert c in O.neighbours(m, True)[:4], (c, k, m)


def test_planar_neighbours():
    assert O.neighbours((0, 1, 0, 0), False) == [
        (O.INVALID,) * 4, (0, 1, 1, 0), (0, 1, 0, 1), (O.INVALID,) * 4,
        (O.INVALID,) * 4, (O.INVALID,) * 4, (0, 1, 1, 1), (O.INVALID,) * 4]
    assert O.children((2, 1, 1, 0)) == [(2, 2, 2, 0), (2, 2, 3, 0), (2, 2, 2, 1), (2, 2, 3, 1)]


def test_cube_seams_are_continuous():
    # A field that is linear in 3D is continuous across cube edges; after stitch every apron pixel
    # must be close to the adjacent own centre pixel.  A wrong projection table flips or transposes
    # the strip and produces jumps of the order of the field's range.
    T, b, lod_count, W = 12, 2, 2, 64
    c = T - 2 * b
    faces = [face_raster(s, W) for s in range(6)]
    a = O.OracleAtlas(lod_count, 64, True, [(T, b, 1, O.FORMAT_R16)])
    a.clear_attachment(0).preprocess_spherical(0, faces, (0, lod_count)).run()
    _, counts = a.task_counts()
    tiles = a.tiles()
    assert len(tiles) == 6 * 5
    o = b + c
    worst = 0
    for (side, lod, x, y), idx in tiles:
        t = a.tile(0, idx).astype(np.int64)
        n = 1 << lod
        step = 2.0 / (n * c) * 16000.0 * 2.5  # generous bound on |grad| * pixel pitch * 2 pixels
        for k in range(b):
            # top/bottom/left/right aprons vs the nearest centre row/column
            for strip, ref in ((t[k, b:o], t[b, b:o]), (t[o + k, b:o], t[o - 1, b:o]),
                               (t[b:o, k], t[b:o, b]), (t[b:o, o + k], t[b:o, o - 1])):
                d = np.abs(strip - ref).max()
                worst = max(worst, d)
                assert d <= step, (side, lod, x, y, k, d, step)
    assert worst > 0  # not a trivially constant field
