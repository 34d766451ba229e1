"""Second, independent models of the parts of the contract the reference expresses in RUST (no executable counterpart here):
written from the reference text a second time, in another shape than bt_host.cpp / bt_oracle*.c — closed forms and whole-table
numpy array operations instead of replayed pushes and per-node loops — so that a misreading shared by the product and the oracle
(same author, same reading) does not pass unnoticed.  Nothing here imports the oracle or the product.

  atlas_indices / planar_closed_form   preprocess/preprocessor.rs:58-66, 234-343 + terrain_data/tile_atlas.rs:383-416
  TileTreeModel                        terrain_data/tile_tree.rs:175-333 + math/coordinate.rs:69-160 + math/terrain_model.rs:130-153
  generate_mipmaps                     terrain_data/mod.rs:143-219
  tc_encode                            formats/mod.rs:8-35 (bincode 2 `config::standard()`: little endian, variable-length integers)
"""
import numpy as np

INVALID = 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ queue order / atlas indices
def overlapping_rect(top_left, bottom_right, lod):
    """PreprocessDataset::overlapping_tiles (preprocessor.rs:58-66): f32 products, `as_uvec2` truncates, `ceil` first for the
    upper corner.  Returns (x0, x1, y0, y1), upper bounds exclusive."""
    tc = np.float32(1 << lod)
    lo = [int(np.float32(v) * tc) for v in top_left]
    hi = [int(np.ceil(np.float32(v) * tc)) for v in bottom_right]
    # (library and oracle clamp the range to the face — tiles with x, y >= 2^lod do not exist; upstream a bottom_right > 1 would
    # invent them: DESIGN.md "deliberate deviations")
    n = 1 << lod
    return min(max(lo[0], 0), n), min(max(hi[0], 0), n), min(max(lo[1], 0), n), min(max(hi[1], 0), n)


def atlas_indices(jobs, first_free=0):
    """jobs: a list of ("tile", side, top_left, bottom_right, lod_begin, lod_end) / ("spherical", lod_begin, lod_end) in call order
    on ONE atlas (tile_states survive clear_attachment: a tile keeps the index of its first appearance).
    An index is handed out the first time get_or_allocate_tile sees a coordinate, counting up from the front of unused_tiles
    (tile_atlas.rs:383-416): the Split tasks of the finest LOD in iproduct order (x outer, y inner), then the Downsample tasks of
    every coarser LOD downwards (preprocessor.rs:234-268); preprocess_spherical does that side after side (:329-331); the Stitch /
    Save tasks that follow only meet coordinates that exist.  Vectorised: every (job, LOD) rectangle as one block of keys, one
    np.unique over the concatenation, ranks by first occurrence.  -> {(side, lod, x, y): atlas_index}"""
    blocks = []
    for job in jobs:
        if job[0] == "tile":
            _, side, tl, br, lod_begin, lod_end = job
            sides = [(side, tl, br)]
        else:
            _, lod_begin, lod_end = job
            sides = [(s, (0.0, 0.0), (1.0, 1.0)) for s in range(6)]
        for side, tl, br in sides:
            for lod in range(lod_end - 1, lod_begin - 1, -1):
                x0, x1, y0, y1 = overlapping_rect(tl, br, lod)
                if x1 <= x0 or y1 <= y0:
                    continue
                xs, ys = np.meshgrid(np.arange(x0, x1, dtype=np.uint64), np.arange(y0, y1, dtype=np.uint64), indexing="ij")  # x outer
                blocks.append((np.uint64(side) << np.uint64(58)) | (np.uint64(lod) << np.uint64(52)) | (xs.ravel() << np.uint64(26)) | ys.ravel())
    if not blocks:
        return {}
    keys = np.concatenate(blocks)
    uniq, first = np.unique(keys, return_index=True)
    order = np.argsort(first, kind="stable")
    out = {}
    for rank, k in enumerate(uniq[order]):
        k = int(k)
        out[(k >> 58, (k >> 52) & 63, (k >> 26) & ((1 << 26) - 1), k & ((1 << 26) - 1))] = first_free + rank
    return out


def planar_closed_form(side, lod, x, y, top_left, bottom_right, lod_begin, lod_end):
    """The atlas index of one tile of ONE preprocess_tile job on a fresh atlas, as arithmetic: every tile of the finer LODs comes
    first (the queue walks lod_range backwards), then the tiles of its own LOD left of its column, then those above it."""
    index = 0
    for l in range(lod_end - 1, lod, -1):
        x0, x1, y0, y1 = overlapping_rect(top_left, bottom_right, l)
        index += max(x1 - x0, 0) * max(y1 - y0, 0)
    x0, x1, y0, y1 = overlapping_rect(top_left, bottom_right, lod)
    assert x0 <= x < x1 and y0 <= y < y1 and lod_begin <= lod < lod_end
    return index + (x - x0) * (y1 - y0) + (y - y0)


# ------------------------------------------------------------------------------------------------ TileTree::update
C_SQR = 0.87 * 0.87  # math/mod.rs:13 (f64)
# math/coordinate.rs:26-42 — 0 = Fixed0, 1 = Fixed1, 2 = PositiveS (u), 3 = PositiveT (v)
EVEN_LIST = [(2, 3), (0, 3), (0, 2), (3, 2), (3, 0), (2, 0)]
ODD_LIST = [(2, 3), (2, 1), (3, 1), (3, 2), (1, 2), (1, 3)]


def _round_half_away(x):
    t = np.trunc(x)
    return t + np.where(np.abs(x - t) >= 0.5, np.sign(x), 0.0)


def _saturating_u32(x):
    return np.clip(np.nan_to_num(x, nan=0.0), 0.0, 4294967295.0).astype(np.uint64).astype(np.uint32)


class TileTreeModel:
    """TileTree::update (tile_tree.rs:268-333) on whole tables.  kind: "planar" (scale = side length) or "sphere" (scale = radius);
    an ellipsoid (scale = (a, b, a)) takes its view coordinate from outside (the projection onto the ellipsoid is not restated here)."""

    def __init__(self, kind, position, scale, min_height, max_height, lod_count, tree_size, load_distance):
        self.kind, self.t = kind, np.asarray(position, np.float64)
        self.scale = np.asarray(scale, np.float64) * np.ones(3)
        self.spherical = kind != "planar"
        self.sides = 6 if self.spherical else 1
        self.lods, self.ts = lod_count, tree_size
        model_scale = float(self.scale[0]) / 2.0 if kind == "planar" else (float(self.scale[0]) if kind == "sphere" else (float(self.scale[0]) + float(self.scale[1])) / 2.0)
        self.load_distance = load_distance * model_scale  # TileTree::new (:144)
        self.approximate_height = np.float32((np.float32(min_height) + np.float32(max_height)) / np.float32(2.0))
        self.heights = (min_height, max_height)
        self.coords = np.full((self.sides, lod_count, tree_size, tree_size, 4), INVALID, np.uint32)
        self.requested = np.zeros((self.sides, lod_count, tree_size, tree_size), bool)
        self.origins = np.zeros((self.sides, lod_count, 2), np.uint32)

    # Coordinate::from_world_position (coordinate.rs:69-107) for the planar and the spherical model
    def view_coordinate(self, p):
        local = (np.asarray(p, np.float64) - self.t) / self.scale
        if not self.spherical:
            return 0, np.clip(np.array([local[0] + 0.5, local[2] + 0.5]), 0.0, 1.0)
        n = local * (1.0 / np.sqrt(local[0] * local[0] + local[1] * local[1] + local[2] * local[2]))
        a = np.abs(n)
        if a[0] > a[1] and a[0] > a[2]:
            side, uv = (0, np.array([-n[2] / n[0], n[1] / n[0]])) if n[0] < 0.0 else (3, np.array([-n[1] / n[0], n[2] / n[0]]))
        elif a[2] > a[1]:
            side, uv = (1, np.array([n[0] / n[2], -n[1] / n[2]])) if n[2] > 0.0 else (4, np.array([n[1] / n[2], -n[0] / n[2]]))
        else:
            side, uv = (2, np.array([n[0] / n[1], n[2] / n[1]])) if n[1] > 0.0 else (5, np.array([-n[2] / n[1], -n[0] / n[1]]))
        w = uv * np.sqrt((1.0 + C_SQR) / (1.0 + C_SQR * uv * uv))
        return side, 0.5 * w + 0.5

    def project_to_side(self, side0, uv, side):  # coordinate.rs:44-52, 136-152
        if not self.spherical:
            return uv
        info = (EVEN_LIST if side0 % 2 == 0 else ODD_LIST)[(6 + side - side0) % 6]
        pick = lambda i: (0.0, 1.0, uv[0], uv[1])[i]
        return np.array([pick(info[0]), pick(info[1])])

    def world_position(self, side, uv, height):  # Coordinate::world_position (:109-131) + position_local_to_world (terrain_model.rs:130-142); uv: (..., 2)
        if self.spherical:
            w = (uv - 0.5) / 0.5
            with np.errstate(invalid="ignore"):  # nodes beyond the face (tree_size > 2^lod): a NaN position, never within the load distance
                q = w / np.sqrt(1.0 + C_SQR - C_SQR * w * w)
            one = np.ones_like(q[..., 0])
            local = [(-one, -q[..., 1], q[..., 0]), (q[..., 0], -q[..., 1], one), (q[..., 0], one, q[..., 1]),
                     (one, -q[..., 0], q[..., 1]), (q[..., 1], -q[..., 0], -one), (q[..., 1], -one, q[..., 0])][side]
            local = np.stack(local, axis=-1)
            local = local * (1.0 / np.sqrt(local[..., 0] * local[..., 0] + local[..., 1] * local[..., 1] + local[..., 2] * local[..., 2]))[..., None]
            direction = local * self.scale
        else:
            local = np.stack([uv[..., 0] - 0.5, np.zeros_like(uv[..., 0]), uv[..., 1] - 0.5], axis=-1)
            direction = np.zeros_like(local) + np.array([0.0, 1.0, 0.0]) * self.scale
        world = local * self.scale + self.t
        normal = direction * (1.0 / np.sqrt(direction[..., 0] * direction[..., 0] + direction[..., 1] * direction[..., 1] + direction[..., 2] * direction[..., 2]))[..., None]
        return world + np.float64(height) * normal

    def update(self, view_position, view_coordinate=None):
        """-> (released, requested): lists of (side, lod, x, y) in the reference's push order (node order)."""
        p = np.asarray(view_position, np.float64)
        side0, uv0 = self.view_coordinate(p) if view_coordinate is None else view_coordinate
        ts = self.ts
        ii, jj = np.meshgrid(np.arange(ts, dtype=np.int64), np.arange(ts, dtype=np.int64), indexing="ij")  # x outer (iproduct!, :282)
        released, requested = [], []
        for side in range(self.sides):
            uv = self.project_to_side(side0, np.asarray(uv0, np.float64), side)
            for lod in range(self.lods):
                tc = float(1 << lod)
                tree_xy = np.minimum(uv * tc, tc - 0.000001)  # compute_tree_xy (:175-178)
                origin = _saturating_u32(np.minimum(np.maximum(_round_half_away(tree_xy - 0.5 * ts), 0.0), tc - ts))  # compute_origin (:180-191)
                self.origins[side, lod] = origin
                tx, ty = int(origin[0]) + ii, int(origin[1]) + jj
                # compute_tile_distance (:193-221)
                off_x, off_y = np.int64(np.trunc(tree_xy[0])) - tx, np.int64(np.trunc(tree_xy[1])) - ty
                frac = np.fmod(tree_xy, 1.0)
                ox = np.where(off_x < 0, 0.0, np.where(off_x > 0, 1.0, frac[0]))
                oy = np.where(off_y < 0, 0.0, np.where(off_y > 0, 1.0, frac[1]))
                tile_uv = np.stack([(tx.astype(np.float64) + ox) / tc, (ty.astype(np.float64) + oy) / tc], axis=-1)
                d = self.world_position(side, tile_uv, self.approximate_height) - p
                distance = np.sqrt(d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1] + d[..., 2] * d[..., 2])
                want = np.full((ts, ts), True) if lod == 0 else distance < self.load_distance / tc
                # the slots (:300-322): a permutation of the ts x ts nodes, so every node meets its own slot exactly once
                sx, sy = tx % ts, ty % ts
                new = np.stack([np.full_like(tx, side), np.full_like(tx, lod), tx, ty], axis=-1).astype(np.uint32)
                old, was = self.coords[side, lod, sx, sy], self.requested[side, lod, sx, sy]
                moved = np.any(old != new, axis=-1)
                release_old = moved & was
                state = was & ~moved
                request = ~state & want
                release_new = state & ~want
                for i, j in zip(ii.ravel(), jj.ravel()):  # node order; only the pushes are walked
                    if release_old[i, j]:
                        released.append(tuple(int(v) for v in old[i, j]))
                    if request[i, j]:
                        requested.append(tuple(int(v) for v in new[i, j]))
                    elif release_new[i, j]:
                        released.append(tuple(int(v) for v in new[i, j]))
                self.coords[side, lod, sx, sy] = new
                self.requested[side, lod, sx, sy] = (state | request) & ~release_new
        return released, requested

    def node_tables(self):
        """(coordinates (nodes, 4), requested (nodes,)) in the library's table order [side][lod][x % ts][y % ts]."""
        return self.coords.reshape(-1, 4).copy(), self.requested.reshape(-1).astype(np.uint32)


# ------------------------------------------------------------------------------------------------ generate_mipmaps
def generate_mipmaps(level0, mip_level_count):
    """AttachmentData::generate_mipmaps (terrain_data/mod.rs:143-219) in integer array arithmetic.  R16 (2-D u16): the mean of the
    NON-ZERO texels of each 2 x 2 block, truncated, 0 when all four are 0; Rgba8 ((T, T, 4) u8): the truncated mean of all four,
    per channel.  -> the mip chain as one flat array, level 0 first (the layout of the reference's Vec)."""
    levels = [np.asarray(level0)]
    for _ in range(1, mip_level_count):
        p = levels[-1].astype(np.uint64)
        blocks = [p[0::2, 0::2], p[0::2, 1::2], p[1::2, 0::2], p[1::2, 1::2]]
        if p.ndim == 2:
            total = sum(blocks)
            count = sum((b != 0).astype(np.uint64) for b in blocks)
            child = np.where(count == 0, 0, total // np.maximum(count, 1))
        else:
            child = sum(blocks) // 4
        levels.append(child.astype(levels[0].dtype))
    return np.concatenate([l.reshape(-1) if l.ndim == 2 else l.reshape(-1, 4) for l in levels])


# ------------------------------------------------------------------------------------------------ TC / bincode
def _varint(v):
    """bincode 2, config::standard(): u < 251 -> one byte; else a marker (251: u16, 252: u32, 253: u64) + little-endian bytes"""
    if v < 251:
        return bytes([v])
    if v < 1 << 16:
        return bytes([251]) + v.to_bytes(2, "little")
    if v < 1 << 32:
        return bytes([252]) + v.to_bytes(4, "little")
    return bytes([253]) + v.to_bytes(8, "little")


def tc_encode(tiles):
    """struct TC { tiles: Vec<TileCoordinate> } (formats/mod.rs:8-11): the length as a usize varint, then side, lod, x, y (u32 varints) per tile"""
    out = bytearray(_varint(len(tiles)))
    for t in tiles:
        for v in t:
            out += _varint(int(v))
    return bytes(out)


# ------------------------------------------------------------------------------------------------ sample_attachment (f4)
def sample_attachment_r16(model, view_position, approximate_height, blend_distance, blend_range, lod_count, entries, layers, T, b, positions):
    """sample_attachment / sample_height (terrain_data/mod.rs:265-307) of an R16 attachment for the planar and the spherical model, one
    sample at a time in the reference's order: surface_position (terrain_model.rs:130-173), compute_blend and lookup_tile
    (tile_tree.rs:223-266), AtlasAttachment::sample (tile_atlas.rs:249-258) and AttachmentData::sample (terrain_data/mod.rs:220-257).
    model: a TileTreeModel (its transform and cube-sphere warp); entries: (sides, lods, tree, tree, 2) u32 = (atlas_index, atlas_lod);
    layers: {atlas_index: (T, T) u16}.  Returns ((n, 4) f32 values, (n,) f32 heights; min / max height from `model.heights`)."""
    f32 = np.float32
    INVALID_LOD = 0xFFFFFFFF
    view = np.asarray(view_position, np.float64)
    scale, offset = f32(T - 2 * b) / f32(T), f32(b) / f32(T)
    ts = entries.shape[2]

    def surface(p):
        local = (np.asarray(p, np.float64) - model.t) / model.scale  # local_from_world
        if model.spherical:
            local = local * (1.0 / np.sqrt(local[0] * local[0] + local[1] * local[1] + local[2] * local[2]))
            direction = local * model.scale
        else:
            local = np.array([1.0, 0.0, 1.0]) * local
            direction = np.array([0.0, 1.0, 0.0]) * model.scale
        world = local * model.scale + model.t
        normal = direction * (1.0 / np.sqrt(direction[0] * direction[0] + direction[1] * direction[1] + direction[2] * direction[2]))
        return world + np.float64(approximate_height) * normal

    def lerp(a, c, t):  # Vec4::lerp / f32::lerp
        return f32(a + f32(f32(c - a) * t))

    def lookup_and_sample(p, lod):
        side, uv = model.view_coordinate(p)
        count = float(1 << lod)
        tree_xy = np.minimum(uv * count, count - 0.000001)
        index, atlas_lod = entries[side, lod, int(tree_xy[0]) % ts, int(tree_xy[1]) % ts]
        if atlas_lod == INVALID_LOD or index == 0xFFFFFFFF:
            return f32(0.0)
        atlas_uv = np.fmod(tree_xy / float(1 << (lod - int(atlas_lod))), 1.0).astype(f32)
        uv32 = atlas_uv * scale + offset
        uvs = uv32 * f32(T) - f32(0.5)
        rem = np.fmod(uvs, f32(1.0))
        ix, iy = int(uvs[0]), int(uvs[1])  # as_ivec2 truncates
        data = layers[int(index)].reshape(-1)
        v = [[f32(data[(iy + y) * T + ix + x]) / f32(65535.0) for y in range(2)] for x in range(2)]
        return lerp(lerp(v[0][0], v[0][1], rem[1]), lerp(v[1][0], v[1][1], rem[1]), rem[0])

    values = np.zeros((len(positions), 4), f32)
    for i, p in enumerate(positions):
        s = surface(p)
        d = s - view
        view_distance = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
        target = f32(min(np.log2(blend_distance / view_distance), float(lod_count) - 0.00001))
        lod = 0 if not target > 0 else int(target)
        value = lookup_and_sample(s, lod)
        if lod != 0:
            a, c = f32(lod) + f32(blend_range), f32(lod)
            ratio = f32(f32(target - a) / f32(c - a))
            ratio = f32(min(max(ratio, f32(0.0)), f32(1.0)))
            if ratio > 0:
                value = lerp(value, lookup_and_sample(s, lod - 1), ratio)
        values[i, 0] = value
    lo, hi = f32(model.heights[0]), f32(model.heights[1])
    return values, (lo + f32(hi - lo) * values[:, 0]).astype(f32)


# ------------------------------------------------------------------------------------------------ streaming state machine
class StreamModel:
    """The streaming half of TileAtlasState (tile_atlas.rs:300-500) as plain Python containers: which atlas slot a requested tile gets
    (the least recently released one), when a slot is taken back, what get_best_tile answers while a tile is still loading."""

    INVALID = 0xFFFFFFFF

    def __init__(self, atlas_size, attachment_count, existing=()):
        from collections import deque

        self.unused = deque((None, i) for i in range(atlas_size))  # (coordinate it last held, atlas index): front = next to go
        self.states = {}  # coordinate -> [requests, attachments still loading, atlas index]
        self.existing = set(existing)
        self.attachments = attachment_count
        self.to_load = deque()

    def _allocate(self):
        coordinate, index = self.unused.popleft()  # ("Atlas out of indices" if empty)
        self.states.pop(coordinate, None)
        return index

    def request_tile(self, c):
        if c not in self.existing:
            return
        st = self.states.get(c)
        if st is not None:
            if st[0] == 0:  # cached but unused: out of the queue again
                self.unused = type(self.unused)(u for u in self.unused if u[1] != st[2])
            st[0] += 1
        else:
            index = self._allocate()
            self.states[c] = [1, self.attachments, index]
            for a in range(self.attachments):
                self.to_load.append((c, index, a))

    def release_tile(self, c):
        if c not in self.existing:
            return
        st = self.states[c]
        st[0] -= 1
        if st[0] == 0:
            self.unused.append((c, st[2]))

    def pending_loads(self):
        return len(self.to_load)

    def finish_loads(self, n):
        done = []
        for _ in range(n):
            c, index, a = self.to_load.popleft()
            st = self.states[c]
            st[1] -= 1
            done.append((c, index))
        return done

    def get_best_tile(self, c):
        side, lod, x, y = c
        while True:
            st = self.states.get((side, lod, x, y))
            if st is not None and st[1] == 0:
                return st[2], lod
            if lod == 0:
                return self.INVALID, self.INVALID
            lod, x, y = lod - 1, x >> 1, y >> 1
