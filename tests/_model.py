"""Independent numpy restatement of the preprocess arithmetic in *mosaic* form.

The reference works tile by tile (split -> downsample -> stitch over an atlas).  SURVEY.md §7.1
claims this equals: build the per-LOD "centre mosaic" pyramid, then cut T x T windows at stride c
with a b-pixel apron.  This module implements that second formulation with vectorised float32
numpy (one IEEE rounding per operation, like the oracle) so the tests can check (a) the oracle
against a structurally different implementation and (b) the equivalence the fused HIP path
relies on.  Test infrastructure only.
"""
from __future__ import annotations

import numpy as np

f32 = np.float32


def _axis_params(size_px, c, n, lo, hi, dim):
    """Per mosaic column/row: clamped texel indices x0,x1 and the fraction (split.wgsl:25-32)."""
    g = np.arange(size_px, dtype=np.uint32)
    tile = (g // c).astype(f32)
    tc = (g % c).astype(f32) / f32(c)
    s = (tile + tc) / f32(n)
    u = (s - f32(lo)) / (f32(hi) - f32(lo))
    q = u * f32(dim) - f32(0.5)
    i = np.floor(q)
    fr = (q - i).astype(f32)
    i = i.astype(np.int64)
    x0 = np.clip(i, 0, dim - 1)
    x1 = np.clip(i + 1, 0, dim - 1)
    return x0, x1, fr


def _mix(a, b, t):
    return (a * (f32(1.0) - t) + b * t).astype(f32)


def _unorm(v, n):
    v = np.clip(v, f32(0.0), f32(1.0)).astype(f32)
    return np.floor(f32(0.5) + f32(n) * v).astype(np.uint32)


def split_mosaic(src, lod, c, top_left=(0.0, 0.0), bottom_right=(1.0, 1.0), x_range=None, y_range=None):
    """Finest-level centre mosaic ((2^lod*c)^2, or the given sub-range) of a source raster.
    src: (H, W) uint16 or (H, W, 4) uint8.  Invalid pixels (any footprint texel ch0 == 0) -> 0."""
    H, W = src.shape[:2]
    n = 1 << lod
    size = n * c
    x0, x1, fx = _axis_params(size, c, n, top_left[0], bottom_right[0], W)
    y0, y1, fy = _axis_params(size, c, n, top_left[1], bottom_right[1], H)
    if x_range is not None:
        x0, x1, fx = x0[x_range[0]:x_range[1]], x1[x_range[0]:x_range[1]], fx[x_range[0]:x_range[1]]
    if y_range is not None:
        y0, y1, fy = y0[y_range[0]:y_range[1]], y1[y_range[0]:y_range[1]], fy[y_range[0]:y_range[1]]
    norm = f32(65535.0) if src.dtype == np.uint16 else f32(255.0)
    t00 = src[np.ix_(y0, x0)]
    t10 = src[np.ix_(y0, x1)]
    t01 = src[np.ix_(y1, x0)]
    t11 = src[np.ix_(y1, x1)]
    ch0 = (lambda t: t) if src.ndim == 2 else (lambda t: t[..., 0])
    valid = (ch0(t00) != 0) & (ch0(t10) != 0) & (ch0(t01) != 0) & (ch0(t11) != 0)
    fxx = fx[None, :] if src.ndim == 2 else fx[None, :, None]
    fyy = fy[:, None] if src.ndim == 2 else fy[:, None, None]
    cvt = lambda t: t.astype(f32) / norm
    top = _mix(cvt(t00), cvt(t10), fxx)
    bot = _mix(cvt(t01), cvt(t11), fxx)
    val = _mix(top, bot, fyy)
    out = _unorm(val, norm).astype(src.dtype)
    if src.ndim == 2:
        out[~valid] = 0
    else:
        out[~valid, :] = 0
    return out


def downsample_mosaic(m):
    """Parent mosaic from a child mosaic (downsample.wgsl:12-40): valid-average of 2x2 blocks in the
    order (0,0),(0,1),(1,0),(1,1) of (dx,dy)."""
    norm = f32(65535.0) if m.dtype == np.uint16 else f32(255.0)
    h, w = m.shape[0] // 2, m.shape[1] // 2
    acc = np.zeros((h, w) + m.shape[2:], dtype=f32)
    cnt = np.zeros((h, w), dtype=f32)
    for dx, dy in ((0, 0), (0, 1), (1, 0), (1, 1)):
        t = m[dy::2, dx::2][:h, :w]
        v = t.astype(f32) / norm
        valid = (t != 0) if m.ndim == 2 else np.any(t[..., :3] != 0, axis=-1)
        if m.ndim == 2:
            acc = np.where(valid, acc + v, acc).astype(f32)
        else:
            acc = np.where(valid[..., None], acc + v, acc).astype(f32)
        cnt = np.where(valid, cnt + f32(1.0), cnt).astype(f32)
    safe = np.where(cnt == 0, f32(1.0), cnt)
    val = acc / (safe if m.ndim == 2 else safe[..., None])
    out = _unorm(val, norm).astype(m.dtype)
    if m.ndim == 2:
        out[cnt == 0] = 0
    else:
        out[cnt == 0, :] = 0
    return out


def build_pyramid(src, lod_count, c, **kw):
    """[mosaic(lod 0), ..., mosaic(lod_count-1)]"""
    levels = [None] * lod_count
    levels[lod_count - 1] = split_mosaic(src, lod_count - 1, c, **kw)
    for lod in range(lod_count - 2, -1, -1):
        levels[lod] = downsample_mosaic(levels[lod + 1])
    return levels


def planar_tile_from_mosaic(m, lod, x, y, T, b):
    """Cut tile (lod,x,y) out of the lod mosaic with the stitch apron rule for a planar terrain whose
    dataset covers the whole [0,1]^2 (stitch.wgsl:53-118): a missing neighbour makes the *whole* region
    clamp into the tile's own centre."""
    c = T - 2 * b
    n = 1 << lod
    out = np.zeros((T, T) + m.shape[2:], dtype=m.dtype)
    o = b + c
    regions = [  # (x, y, w, h), neighbour offset
        ((b, 0, c, b), (0, -1)), ((o, b, b, c), (1, 0)), ((b, o, c, b), (0, 1)), ((0, b, b, c), (-1, 0)),
        ((0, 0, b, b), (-1, -1)), ((o, 0, b, b), (1, -1)), ((o, o, b, b), (1, 1)), ((0, o, b, b), (-1, 1)),
    ]
    gx0, gy0 = x * c, y * c
    out[b:o, b:o] = m[gy0:gy0 + c, gx0:gx0 + c]
    for (rx, ry, rw, rh), (dx, dy) in regions:
        px = np.arange(rx, rx + rw)
        py = np.arange(ry, ry + rh)
        nx, ny = x + dx, y + dy
        if 0 <= nx < n and 0 <= ny < n:
            gx = gx0 + px - b
            gy = gy0 + py - b
        else:
            gx = gx0 + np.clip(px, b, o - 1) - b
            gy = gy0 + np.clip(py, b, o - 1) - b
        out[ry:ry + rh, rx:rx + rw] = m[np.ix_(gy, gx)]
    return out


def fbm_u16(w, h, seed, octaves=6, x0=0, y0=0, base_cell=None):
    """Integer-only value-noise fBm heightmap in [1, 65535] (0 is the no-data sentinel).  Every step
    is exact integer arithmetic so the HIP generator (bt_synth_fbm_r16) reproduces it bit for bit."""
    base_cell = base_cell or max(w, h) // 4 or 1
    xs = (np.arange(w, dtype=np.uint64) + np.uint64(x0))[None, :]
    ys = (np.arange(h, dtype=np.uint64) + np.uint64(y0))[:, None]
    total = np.zeros((h, w), dtype=np.uint64)
    amp_total = 0
    cell = base_cell
    for o in range(octaves):
        cell = max(cell, 1)
        amp = 1 << (octaves - 1 - o)
        total += _value_noise(xs, ys, cell, seed + 0x9E3779B9 * (o + 1)) * np.uint64(amp)
        amp_total += amp
        cell //= 2
    v = total // np.uint64(amp_total)  # 0..65535
    return (np.uint64(1) + (v * np.uint64(65534)) // np.uint64(65535)).astype(np.uint16)


def _hash2(ix, iy, seed):
    m = np.uint64(0xFFFFFFFF)
    h = (ix * np.uint64(0x85EBCA6B) + iy * np.uint64(0xC2B2AE35) + np.uint64(seed & 0xFFFFFFFF)) & m
    h ^= h >> np.uint64(15)
    h = (h * np.uint64(0x2C1B3C6D)) & m
    h ^= h >> np.uint64(12)
    h = (h * np.uint64(0x297A2D39)) & m
    h ^= h >> np.uint64(15)
    return h & np.uint64(0xFFFF)


def _value_noise(xs, ys, cell, seed):
    c = np.uint64(cell)
    ix, fx = xs // c, xs % c
    iy, fy = ys // c, ys % c
    # 12-bit fixed-point smoothstep-free linear weights keep everything in 64-bit integers
    wx = (fx * np.uint64(4096)) // c
    wy = (fy * np.uint64(4096)) // c
    one = np.uint64(4096)
    v00 = _hash2(ix, iy, seed)
    v10 = _hash2(ix + np.uint64(1), iy, seed)
    v01 = _hash2(ix, iy + np.uint64(1), seed)
    v11 = _hash2(ix + np.uint64(1), iy + np.uint64(1), seed)
    top = v00 * (one - wx) + v10 * wx
    bot = v01 * (one - wx) + v11 * wx
    return (top * (one - wy) + bot * wy) >> np.uint64(24)
