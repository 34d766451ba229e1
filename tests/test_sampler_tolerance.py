"""The stated float tolerance of `split` (north_star: "within stated float tolerance (downsampled heights)").

The reference filters the source raster with the GPU's fixed-function sampler (split.wgsl:32), whose bilinear weights
are implementation-defined fixed-point numbers (>= 8 fractional bits on every desktop GPU); the oracle and the product
define the filter as exact f32 `mix`.  The oracle's sampler model (orc_set_sampler_model) restates the hardware
behaviour; these tests bound the distance between the two definitions:

    |exact - N-bit-weights|  <=  2 * eps_N * D + 1   [LSB of the attachment format]

with eps_N the largest weight error (2^-(N+1) to nearest, 2^-N truncated) and D the largest difference between
horizontally / vertically adjacent source texels (LSB): each axis moves the sample by at most eps_N * D, the final
quantisation adds at most one more LSB.  Coarser LODs average 2 x 2 blocks, which cannot increase the bound.
"""
import json
import os

import numpy as np
import pytest

import _cases as K
import _oracle as O


def adjacent_contrast(src):
    s = src.astype(np.int32)
    if s.ndim == 2:
        s = s[..., None]
    return int(max(np.abs(np.diff(s, axis=0)).max(), np.abs(np.diff(s, axis=1)).max()))


def per_lod_max_diff(tiles, a, b, attachment=0):
    """a(idx), b(idx) -> texel arrays; returns {lod: max |a - b| in LSB}."""
    out = {}
    for coord, idx in tiles:
        d = int(np.abs(a(idx).astype(np.int64) - b(idx).astype(np.int64)).max())
        out[coord[1]] = max(out.get(coord[1], 0), d)
    return out


@pytest.mark.parametrize("bits,mode", [(8, 0), (8, 1), (6, 0)])
@pytest.mark.parametrize("fmt,smooth", [(O.FORMAT_R16, True), (O.FORMAT_R16, False), (O.FORMAT_RGBA8, False)])
def test_weight_quantisation_bound_small(bits, mode, fmt, smooth):
    T, b, lods = 64, 2, 3
    src = K.smooth_raster(300, 300, seed=5) if smooth else K.random_raster(fmt, 300, 300, seed=5)
    exact = K.oracle_planar(src, lods, T, b, fmt)
    with O.sampler_model(bits, mode):
        snapped = K.oracle_planar(src, lods, T, b, fmt)
    eps = 2.0 ** -(bits + 1) if mode == 0 else 2.0 ** -bits
    bound = 2 * eps * adjacent_contrast(src) + 1
    diffs = per_lod_max_diff(exact.tiles(), lambda i: exact.tile(0, i), lambda i: snapped.tile(0, i))
    assert max(diffs.values()) <= bound, (diffs, bound)
    assert max(diffs.values()) > 0 or smooth  # the model does change results on a contrasty raster
    assert diffs[0] <= diffs[lods - 1] + 1  # coarser LODs are averages of the finest one


def test_exact_model_is_the_default():
    src = K.random_raster(O.FORMAT_R16, 100, 100, seed=2)
    a = K.oracle_planar(src, 2, 32, 2, O.FORMAT_R16)
    with O.sampler_model(8):
        pass
    b = K.oracle_planar(src, 2, 32, 2, O.FORMAT_R16)
    assert all(np.array_equal(a.tile(0, i), b.tile(0, i)) for _, i in a.tiles())


def _report(name, payload):
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, f"sampler_tolerance_{name}.json"), "w") as f:
        json.dump(payload, f, indent=1)


@pytest.mark.gpu
def test_config2_product_vs_8bit_sampler():
    """BASELINE config 2 (4096^2 height R16 + albedo Rgba8, lod_count 4): product (HIP, exact f32 filter) against the
    oracle with 8-bit sampler weights, per LOD, in LSB."""
    import bevy_terrain_amd as bt

    device = bt.Device(0)
    height = K.smooth_raster(4096, 4096, seed=1234, device=device)
    rng = np.random.default_rng(1235)
    albedo = rng.integers(1, 256, size=(4096, 4096, 4), dtype=np.uint8)
    albedo[..., 3] = 255
    report = {}
    for name, src, fmt in (("height", height, O.FORMAT_R16), ("albedo", albedo, O.FORMAT_RGBA8)):
        atlas, _ = K.product_planar(device, src, 4, 512, 2, fmt, atlas_size=128)
        with O.sampler_model(8, 0):
            snapped = K.oracle_planar(src, 4, 512, 2, fmt, atlas_size=128, threads=O.usable_cores())
        ours = atlas.download_tiles(0, 0, 85)
        diffs = per_lod_max_diff(snapped.tiles(), lambda i: ours[i], lambda i: snapped.tile(0, i))
        contrast = adjacent_contrast(src)
        bound = 2 * 2.0 ** -9 * contrast + 1
        report[name] = {"max_abs_diff_lsb_per_lod": diffs, "adjacent_contrast_lsb": contrast, "bound_lsb": bound}
        assert max(diffs.values()) <= bound, report
    _report("config2", report)


@pytest.mark.gpu
def test_config3_product_vs_8bit_sampler():
    """BASELINE config 3 (16384^2 fBm, lod_count 6, 1365 tiles): the same measurement at full size."""
    import bevy_terrain_amd as bt

    device = bt.Device(0)
    src = K.smooth_raster(16384, 16384, seed=42, device=device)
    atlas, _ = K.product_planar(device, src, 6, 512, 2, O.FORMAT_R16, atlas_size=2048)
    with O.sampler_model(8, 0):
        snapped = K.oracle_planar(src, 6, 512, 2, O.FORMAT_R16, atlas_size=2048, threads=O.usable_cores())
    diffs = {}
    tiles = snapped.tiles()
    for first in range(0, 1365, 128):
        count = min(128, 1365 - first)
        ours = atlas.download_tiles(0, first, count)
        part = per_lod_max_diff(tiles[first:first + count], lambda i: ours[i - first], lambda i: snapped.tile(0, i))
        for lod, d in part.items():
            diffs[lod] = max(diffs.get(lod, 0), d)
    contrast = adjacent_contrast(src)
    bound = 2 * 2.0 ** -9 * contrast + 1
    _report("config3", {"height": {"max_abs_diff_lsb_per_lod": diffs, "adjacent_contrast_lsb": contrast, "bound_lsb": bound}})
    assert max(diffs.values()) <= bound, (diffs, bound)
