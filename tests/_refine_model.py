"""An independent second model of the tiling prepass (the first one is oracle/bt_oracle.c's statement-by-statement
restatement): whole passes as numpy float32 array operations, breadth first.  TEST INFRASTRUCTURE ONLY.

The reference's ping-pong buffer (prepare_prepass.wgsl:25-36) makes every pass read the previous pass's children in
append order, so with invocations taken in id order the prepass is a breadth-first traversal: per pass, the tiles that
do not divide are appended to the final list in order, the others contribute their four children in order
(refine_tiles.wgsl:24-44).  The distance test is functions.wgsl:73-96,117-188 on arrays; numpy's float32 +, -, *, /,
sqrt are IEEE, like the shader's (no fused operations)."""
import numpy as np

F = np.float32


def _change_lod(xy, uv, lod_from, lod_to):
    """coordinate_change_lod (functions.wgsl:164-188) of ONE coordinate to an array of target lods."""
    d = lod_to.astype(np.int64) - int(lod_from)
    out_xy = np.empty((len(d), 2), np.uint32)
    out_uv = np.empty((len(d), 2), F)
    up = d > 0
    delta_count = (np.uint64(1) << np.abs(d).astype(np.uint64)).astype(np.uint64)
    delta_size = np.ldexp(F(1.0), d.astype(np.int32)).astype(F)
    for k in range(2):
        scaled = (F(uv[k]) * delta_size).astype(F)
        out_xy[:, k] = np.where(up, (np.uint64(xy[k]) * delta_count + np.trunc(scaled).astype(np.uint64)) & np.uint64(0xFFFFFFFF),
                                np.uint64(xy[k]) // delta_count).astype(np.uint32)
        down_uv = ((np.uint64(xy[k]) % delta_count).astype(F) + F(uv[k])).astype(F) * delta_size
        out_uv[:, k] = np.where(up, scaled - np.trunc(scaled), down_uv).astype(F)
    same = d == 0
    out_xy[same] = np.asarray(xy, np.uint32)
    out_uv[same] = np.asarray(uv, F)
    return out_xy, out_uv


def should_be_divided(view, tiles):
    """tiles: (n, 4) uint32 [side, lod, x, y] -> (divide mask, view distance)."""
    n = len(tiles)
    side, lod, x, y = (tiles[:, k] for k in range(4))
    vx = np.empty((n, 2), np.uint32)
    vuv = np.empty((n, 2), F)
    for s in np.unique(side):
        m = side == s
        p = view.sides[int(s)]
        vx[m], vuv[m] = _change_lod((p.view_xy[0], p.view_xy[1]), (p.view_uv[0], p.view_uv[1]), view.origin_lod, lod[m])
    off_x = vx[:, 0].astype(np.int64) - x.astype(np.int64)
    off_y = vx[:, 1].astype(np.int64) - y.astype(np.int64)
    u = np.where(off_x < 0, F(0), np.where(off_x > 0, F(1), vuv[:, 0])).astype(F)
    w = np.where(off_y < 0, F(0), np.where(off_y > 0, F(1), vuv[:, 1])).astype(F)
    tc = np.ldexp(F(1.0), lod.astype(np.int32)).astype(F)
    u = ((x.astype(F) + u) / tc).astype(F)
    w = ((y.astype(F) + w) / tc).astype(F)
    if view.spherical:
        c = F(0.87) * F(0.87)
        u = (u - F(0.5)) / F(0.5)
        w = (w - F(0.5)) / F(0.5)
        u = u / np.sqrt(F(1.0) + c - c * u * u)
        w = w / np.sqrt(F(1.0) + c - c * w * w)
        one = np.ones(n, F)
        faces = {0: (-one, -w, u), 1: (u, -w, one), 2: (u, one, w), 3: (one, -u, w), 4: (w, -u, -one), 5: (w, -one, u)}
        l = np.zeros((n, 3), F)
        for s, (a, b, cc) in faces.items():
            m = side == s
            l[m, 0], l[m, 1], l[m, 2] = a[m], b[m], cc[m]
        ln = np.sqrt(l[:, 0] * l[:, 0] + l[:, 1] * l[:, 1] + l[:, 2] * l[:, 2])
        l = (l / ln[:, None]).astype(F)
        normal0 = l
    else:
        l = np.stack([u - F(0.5), np.zeros(n, F), w - F(0.5)], axis=1).astype(F)
        normal0 = np.tile(np.array([0, 1, 0], F), (n, 1))
    m = np.array(list(view.world_from_local), F)
    world = np.stack([(m[r] * l[:, 0] + m[3 + r] * l[:, 1] + m[6 + r] * l[:, 2]) + m[9 + r] for r in range(3)], axis=1).astype(F)
    t = np.array(list(view.local_from_world_transpose), F)
    nrm = np.stack([t[r] * normal0[:, 0] + t[3 + r] * normal0[:, 1] + t[6 + r] * normal0[:, 2] for r in range(3)], axis=1).astype(F)
    nl = np.sqrt(nrm[:, 0] * nrm[:, 0] + nrm[:, 1] * nrm[:, 1] + nrm[:, 2] * nrm[:, 2])
    nrm = (nrm / nl[:, None]).astype(F)
    wp = np.array(list(view.world_position), F)
    d = (world + F(view.approximate_height) * nrm) - wp
    dist = np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(F)
    return dist < (F(view.subdivision_distance) / tc), dist


def refine(view):
    """-> (final tiles (n, 4) uint32 in append order, tiles dropped on the last pass, per-pass tile counts)."""
    tile_count = getattr(view, "tile_count", None) or view.geometry_tile_count
    roots = 6 if view.spherical else 1
    current = np.array([[s, 0, 0, 0] for s in range(roots)], np.uint32)
    final, counts = [], []
    dropped = np.zeros((0, 4), np.uint32)
    for p in range(view.refinement_count + 1):
        counts.append(len(current))
        if len(current) == 0:
            continue
        divide, _ = should_be_divided(view, current)
        final.append(current[~divide])
        parents = current[divide]
        if p == view.refinement_count:
            dropped = parents
            break
        i = np.tile(np.arange(4, dtype=np.uint32), len(parents))
        rep = np.repeat(parents, 4, axis=0)
        current = np.stack([rep[:, 0], rep[:, 1] + 1, (rep[:, 2] << 1) + (i & 1), (rep[:, 3] << 1) + ((i >> 1) & 1)], axis=1).astype(np.uint32)
        assert len(current) <= tile_count, "temporary_tiles overflow"
    final = np.concatenate(final) if final else np.zeros((0, 4), np.uint32)
    return final, dropped, counts


def check_quadtree(final, dropped, roots):
    """Quadtree invariants of a prepass result: tiles pairwise disjoint; final + dropped tiles cover every root
    exactly (area sum 4^-lod per side == 1); returns the largest LOD difference between edge-adjacent final tiles."""
    every = np.concatenate([final, dropped]) if len(dropped) else final
    s = {tuple(t) for t in every.tolist()}
    assert len(s) == len(every), "a tile appears twice"
    for side, lod, x, y in s:
        l, xx, yy = lod, x, y
        while l > 0:
            l, xx, yy = l - 1, xx >> 1, yy >> 1
            assert (side, l, xx, yy) not in s, "a tile and its ancestor are both present"
    from fractions import Fraction
    for side in range(roots):
        area = sum(Fraction(1, 4 ** int(t[1])) for t in every if t[0] == side)
        assert area == 1, (side, area)
    # neighbour LOD difference (same side): rasterise the final tiles' lod at the finest resolution
    worst = 0
    if len(final):
        top = int(min(final[:, 1].max(), 11))
        n = 1 << top
        for side in range(roots):
            grid = np.full((n, n), -1, np.int64)
            for _, lod, x, y in final[final[:, 0] == side]:
                lod = int(lod)
                if lod > top:
                    grid[int(x) >> (lod - top), int(y) >> (lod - top)] = np.maximum(grid[int(x) >> (lod - top), int(y) >> (lod - top)], lod)
                else:
                    k = 1 << (top - lod)
                    grid[int(x) * k:(int(x) + 1) * k, int(y) * k:(int(y) + 1) * k] = lod
            for a, b in ((grid[1:, :], grid[:-1, :]), (grid[:, 1:], grid[:, :-1])):
                ok = (a >= 0) & (b >= 0)
                if ok.any():
                    worst = max(worst, int(np.abs(a[ok] - b[ok]).max()))
    return worst
