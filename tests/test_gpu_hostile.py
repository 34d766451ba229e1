"""Hostile arguments through the C ABI with real objects on the GPU: configurations, rasters, dataset rectangles and ranges no sane host
passes.  The contract: an error status (with a message) or a completed run — never a crash, a hang, or gigabytes of host memory allocated
on the way to the error.  What is refused and what runs is pinned here case by case (round 6 sweep, profiles/r06_hostile_inputs.txt)."""
import ctypes as C

import numpy as np
import pytest

from bevy_terrain_amd import _ffi

pytestmark = pytest.mark.gpu


def cfg(lod_count=3, atlas_size=64, spherical=0, atts=((64, 2, 1, 1),), path=b"terrains/h"):
    c = _ffi.TerrainConfigC()
    c.lod_count, c.atlas_size, c.spherical, c.attachment_count = lod_count, atlas_size, spherical, len(atts)
    for i, (T, b, mips, fmt) in enumerate(atts[:8]):
        c.attachments[i].name = b"att%d" % i
        c.attachments[i].texture_size, c.attachments[i].border_size, c.attachments[i].mip_level_count, c.attachments[i].format = T, b, mips, fmt
    c.path = path
    return c

def objects(L, config):
    ctx = C.c_void_p()
    assert L.bt_ctx_create(0, None, C.byref(ctx)) == 0
    atlas = C.c_void_p()
    s = L.bt_atlas_create(ctx, C.byref(config), C.byref(atlas))
    return ctx, atlas, s

def raster(arr, fmt=1, w=None, h=None, pitch=None, on_device=0):
    r = _ffi.RasterC()
    r.data = arr.ctypes.data if arr is not None else None
    r.width = arr.shape[1] if w is None else w
    r.height = arr.shape[0] if h is None else h
    r.row_pitch = (arr.strides[0] if arr is not None else 0) if pitch is None else pitch
    r.format, r.on_device = fmt, on_device
    return r

def dataset(ai=0, side=0, tl=(0.0, 0.0), br=(1.0, 1.0), lods=(0, 3)):
    d = _ffi.PreprocessDatasetC()
    d.attachment_index, d.side = ai, side
    d.top_left[0], d.top_left[1], d.bottom_right[0], d.bottom_right[1] = tl[0], tl[1], br[0], br[1]
    d.lod_begin, d.lod_end = lods
    return d

def job(L, config, ds, arr=None, rast=None, run_flags=0, clear=True):
    ctx, atlas, s = objects(L, config)
    pre = C.c_void_p()
    try:
        if s != 0:
            return "atlas_create %d %s" % (s, L.bt_last_error().decode()[:80])
        assert L.bt_preprocessor_create(ctx, C.byref(pre)) == 0
        if clear:
            s = L.bt_preprocessor_clear_attachment(pre, atlas, ds.attachment_index, None)
            if s != 0:
                return "clear %d %s" % (s, L.bt_last_error().decode()[:80])
        if arr is None and rast is None:
            arr = u16(200, 200)
        r = rast if rast is not None else raster(arr)
        s = L.bt_preprocessor_preprocess_tile(pre, atlas, C.byref(ds), C.byref(r))
        if s != 0:
            return "preprocess_tile %d %s" % (s, L.bt_last_error().decode()[:80])
        s = L.bt_preprocessor_run(pre, atlas, run_flags)
        if s != 0:
            return "run %d %s" % (s, L.bt_last_error().decode()[:80])
        return "ok sync %d" % L.bt_ctx_synchronize(ctx)
    finally:
        L.bt_ctx_synchronize(ctx)
        L.bt_preprocessor_destroy(pre)
        L.bt_atlas_destroy(atlas)
        L.bt_ctx_destroy(ctx)


u16 = lambda h, w: (np.arange(h * w, dtype=np.uint32) % 65535 + 1).astype(np.uint16).reshape(h, w)

CASES = [
    ("lod_count 0", lambda L: job(L, cfg(lod_count=0), dataset(lods=(0, 0)))),
    ("lod_count 0, lods (0,3)", lambda L: job(L, cfg(lod_count=0), dataset(lods=(0, 3)))),
    ("lod_count 33", lambda L: job(L, cfg(lod_count=33, atlas_size=64), dataset(lods=(0, 2)))),
    ("lod range beyond lod_count", lambda L: job(L, cfg(lod_count=2), dataset(lods=(0, 5)))),
    ("lod range inverted", lambda L: job(L, cfg(), dataset(lods=(3, 1)))),
    ("lod range (31,40)", lambda L: job(L, cfg(lod_count=3), dataset(lods=(31, 40)))),
    ("atlas_size 0", lambda L: job(L, cfg(atlas_size=0), dataset())),
    ("atlas_size 1", lambda L: job(L, cfg(atlas_size=1), dataset())),
    ("atlas_size 0xFFFFFFFF", lambda L: job(L, cfg(atlas_size=0xFFFFFFFF), dataset())),
    ("no attachments", lambda L: job(L, cfg(atts=()), dataset())),
    ("attachment index 5", lambda L: job(L, cfg(), dataset(ai=5))),
    ("T 0", lambda L: job(L, cfg(atts=((0, 0, 1, 1),)), dataset())),
    ("T 1 b 0", lambda L: job(L, cfg(atts=((1, 0, 1, 1),)), dataset())),
    ("T 2 b 0", lambda L: job(L, cfg(atts=((2, 0, 1, 1),)), dataset())),
    ("T 3 b 1", lambda L: job(L, cfg(atts=((3, 1, 1, 1),)), dataset())),
    ("T 5 b 1", lambda L: job(L, cfg(atts=((5, 1, 1, 1),)), dataset())),
    ("T 7 b 0", lambda L: job(L, cfg(atts=((7, 0, 1, 1),)), dataset())),
    ("T 63 b 2 (odd)", lambda L: job(L, cfg(atts=((63, 2, 1, 1),)), dataset())),
    ("T 64 b 31", lambda L: job(L, cfg(atts=((64, 31, 1, 1),)), dataset())),
    ("T 64 b 32", lambda L: job(L, cfg(atts=((64, 32, 1, 1),)), dataset())),
    ("T 64 b 0", lambda L: job(L, cfg(atts=((64, 0, 1, 1),)), dataset())),
    ("T 64 b 17 (b > c/2)", lambda L: job(L, cfg(atts=((64, 17, 1, 1),)), dataset())),
    ("T 4096 b 2", lambda L: job(L, cfg(atlas_size=8, lod_count=1, atts=((4096, 2, 1, 1),)), dataset(lods=(0, 1)))),
    ("mips 0", lambda L: job(L, cfg(atts=((64, 2, 0, 1),)), dataset())),
    ("mips 7 of a 64-texel tile", lambda L: job(L, cfg(atts=((64, 2, 7, 1),)), dataset())),
    ("mips 8 of a 64-texel tile", lambda L: job(L, cfg(atts=((64, 2, 8, 1),)), dataset())),
    ("mips 0xFFFFFFFF", lambda L: job(L, cfg(atts=((64, 2, 0xFFFFFFFF, 1),)), dataset())),
    ("format 7", lambda L: job(L, cfg(atts=((64, 2, 1, 7),)), dataset())),
    ("format Rg16", lambda L: job(L, cfg(atts=((64, 2, 1, 3),)), dataset())),
    ("format Rgb8", lambda L: job(L, cfg(atts=((64, 2, 1, 5),)), dataset())),
    ("raster format != attachment format", lambda L: job(L, cfg(), dataset(), rast=raster(u16(100, 100), fmt=0))),
    ("raster data NULL", lambda L: job(L, cfg(), dataset(), rast=raster(None, w=10, h=10, pitch=20))),
    ("raster 0 x 0", lambda L: job(L, cfg(), dataset(), rast=raster(u16(4, 4), w=0, h=0))),
    ("raster 1 x 1", lambda L: job(L, cfg(), dataset(), arr=u16(1, 1))),
    ("raster 1 x 3000", lambda L: job(L, cfg(), dataset(), arr=u16(3000, 1))),
    ("raster 3000 x 1", lambda L: job(L, cfg(), dataset(), arr=u16(1, 3000))),
    ("raster pitch < width", lambda L: job(L, cfg(), dataset(), rast=raster(u16(100, 100), pitch=50))),
    ("raster pitch odd", lambda L: job(L, cfg(), dataset(), rast=raster(u16(100, 101), w=100, pitch=201))),
    ("raster pitch 0", lambda L: job(L, cfg(), dataset(), rast=raster(u16(100, 100), pitch=0))),
    ("raster on_device 7", lambda L: job(L, cfg(), dataset(), rast=raster(u16(100, 100), on_device=7))),
    ("rect inverted", lambda L: job(L, cfg(), dataset(tl=(0.8, 0.8), br=(0.2, 0.2)))),
    ("rect zero area", lambda L: job(L, cfg(), dataset(tl=(0.5, 0.5), br=(0.5, 0.5)))),
    ("rect outside", lambda L: job(L, cfg(), dataset(tl=(1.5, 1.5), br=(2.5, 2.5)))),
    ("rect negative", lambda L: job(L, cfg(), dataset(tl=(-1.0, -1.0), br=(0.5, 0.5)))),
    ("rect huge", lambda L: job(L, cfg(), dataset(tl=(-1e30, -1e30), br=(1e30, 1e30)))),
    ("rect nan", lambda L: job(L, cfg(), dataset(tl=(float("nan"), 0.0), br=(1.0, float("nan"))))),
    ("rect inf", lambda L: job(L, cfg(), dataset(tl=(float("-inf"), 0.0), br=(float("inf"), 1.0)))),
    ("rect tiny", lambda L: job(L, cfg(), dataset(tl=(0.5, 0.5), br=(0.5 + 1e-7, 0.5 + 1e-7)))),
    ("side 9 planar", lambda L: job(L, cfg(), dataset(side=9))),
    ("side 9 spherical", lambda L: job(L, cfg(spherical=1, atlas_size=256), dataset(side=9))),
    ("side 5 spherical only", lambda L: job(L, cfg(spherical=1, atlas_size=256), dataset(side=5))),
    ("atlas too small (overflow)", lambda L: job(L, cfg(atlas_size=3), dataset())),
    ("run generic", lambda L: job(L, cfg(), dataset(), run_flags=1)),
    ("run all flags", lambda L: job(L, cfg(), dataset(), run_flags=0xFFFFFFFF)),
    ("run reference dispatch T 20", lambda L: job(L, cfg(atts=((20, 2, 1, 1),)), dataset(), run_flags=256)),
    ("no clear", lambda L: job(L, cfg(), dataset(), clear=False)),
    ("lod 12 tiny tiles", lambda L: job(L, cfg(lod_count=12, atlas_size=64, atts=((12, 2, 1, 1),)), dataset(lods=(0, 12)))),
]


EXPECTED = {'lod_count 0': 'preprocess_tile -1',
            'lod_count 0, lods (0,3)': 'ok',
            'lod_count 33': 'ok',
            'lod range beyond lod_count': 'preprocess_tile -2',
            'lod range inverted': 'preprocess_tile -1',
            'lod range (31,40)': 'preprocess_tile -1',
            'atlas_size 0': 'preprocess_tile -2',
            'atlas_size 1': 'preprocess_tile -2',
            'atlas_size 0xFFFFFFFF': 'atlas_create -3',
            'no attachments': 'clear -1',
            'attachment index 5': 'clear -1',
            'T 0': 'atlas_create -1',
            'T 1 b 0': 'preprocess_tile -5',
            'T 2 b 0': 'ok',
            'T 3 b 1': 'preprocess_tile -5',
            'T 5 b 1': 'preprocess_tile -5',
            'T 7 b 0': 'preprocess_tile -5',
            'T 63 b 2 (odd)': 'preprocess_tile -5',
            'T 64 b 31': 'preprocess_tile -5',
            'T 64 b 32': 'atlas_create -1',
            'T 64 b 0': 'ok',
            'T 64 b 17 (b > c/2)': 'ok',
            'T 4096 b 2': 'ok',
            'mips 0': 'ok',
            'mips 7 of a 64-texel tile': 'ok',
            'mips 8 of a 64-texel tile': 'atlas_create -1',
            'mips 0xFFFFFFFF': 'atlas_create -1',
            'format 7': 'preprocess_tile -5',
            'format Rg16': 'preprocess_tile -5',
            'format Rgb8': 'preprocess_tile -5',
            'raster format != attachment format': 'preprocess_tile -1',
            'raster data NULL': 'preprocess_tile -1',
            'raster 0 x 0': 'preprocess_tile -1',
            'raster 1 x 1': 'ok',
            'raster 1 x 3000': 'ok',
            'raster 3000 x 1': 'ok',
            'raster pitch < width': 'preprocess_tile -1',
            'raster pitch odd': 'preprocess_tile -1',
            'raster pitch 0': 'ok',
            'raster on_device 7': 'preprocess_tile -1',
            'rect inverted': 'preprocess_tile -1',
            'rect zero area': 'preprocess_tile -1',
            'rect outside': 'ok',
            'rect negative': 'ok',
            'rect huge': 'ok',
            'rect nan': 'preprocess_tile -1',
            'rect inf': 'ok',
            'rect tiny': 'ok',
            'side 9 planar': 'preprocess_tile -1',
            'side 9 spherical': 'preprocess_tile -1',
            'side 5 spherical only': 'ok',
            'atlas too small (overflow)': 'preprocess_tile -2',
            'run generic': 'ok',
            'run all flags': 'ok',
            'run reference dispatch T 20': 'ok',
            'no clear': 'ok',
            'lod 12 tiny tiles': 'preprocess_tile -2'}


@pytest.mark.parametrize("name", [c[0] for c in CASES])
def test_hostile_job(name):
    fn = dict(CASES)[name]
    result = fn(_ffi.lib())
    assert result.startswith(EXPECTED[name]), (name, result)


def test_nine_attachments_are_refused():
    c = cfg()
    c.attachment_count = 9
    assert objects(_ffi.lib(), c)[2] == -1


def test_calls_on_an_empty_queue_and_a_foreign_atlas(tmp_path):
    L = _ffi.lib()
    ctx, atlas, s = objects(L, cfg())
    pre = C.c_void_p()
    assert L.bt_preprocessor_create(ctx, C.byref(pre)) == 0
    root = str(tmp_path).encode()
    assert [L.bt_preprocessor_run(pre, atlas, 0), L.bt_preprocessor_run(pre, atlas, 0), L.bt_preprocessor_save(pre, atlas, root),
            L.bt_preprocessor_run_streamed(pre, atlas, root, 0, None)] == [0, 0, 0, 0]
    ctx2, atlas2, s2 = objects(L, cfg(atts=((32, 1, 1, 0),)))
    ds, arr = dataset(), u16(100, 100)
    r = raster(arr)
    assert L.bt_preprocessor_clear_attachment(pre, atlas, 0, None) == 0 and L.bt_preprocessor_preprocess_tile(pre, atlas, C.byref(ds), C.byref(r)) == 0
    assert L.bt_preprocessor_run(pre, atlas2, 0) == -1 and b"different contexts" in L.bt_last_error()


def test_io_and_range_errors_are_statuses():
    L = _ffi.lib()
    ctx, atlas, s = objects(L, cfg())
    assert L.bt_atlas_save_attachment(atlas, 0, b"/proc/nope/x") == -4
    assert L.bt_atlas_load_tile_config(atlas, b"/nonexistent/config.tc") == -4
    assert L.bt_atlas_load_tiles(atlas, 0, b"/nonexistent", None, 0) == 0  # (nothing exists: nothing to load)
    assert L.bt_atlas_download_tiles(atlas, 0, 60, 10, None, 0) == -1
    assert L.bt_atlas_generate_mipmaps(atlas, 0, 60, 10) == -1 and L.bt_atlas_generate_mipmaps(atlas, 3, 0, 1) == -1


def test_tile_tree_and_prepass_sizes():
    L = _ffi.lib()
    ctx, atlas, s = objects(L, cfg())
    got = []
    for tree_size, lods in ((0, 3), (1, 3), (3, 3), (1 << 20, 3), (8, 0), (8, 40)):
        vc = _ffi.TerrainViewConfigC()
        L.bt_terrain_view_config_default(C.byref(vc))
        vc.tree_size = tree_size
        m = _ffi.TerrainModelC()
        m.kind, m.a, m.b, m.max_height = 0, 1000.0, 1000.0, 1.0
        t = C.c_void_p()
        st = L.bt_tile_tree_create(ctx, C.byref(m), lods, C.byref(vc), C.byref(t))
        if st == 0:
            st = (st, L.bt_tile_tree_update(t, (C.c_double * 3)(1.0, 2.0, 3.0)))
        got.append(st)
    assert got == [-1, (0, 0), (0, 0), -1, -1, -1]
    got = []
    for count in (0, 1, 0xFFFFFFFF):
        pp = C.c_void_p()
        got.append(L.bt_tiling_prepass_create(ctx, count, C.byref(pp)))
        L.bt_tiling_prepass_destroy(pp)
    assert got[:2] == [-1, -1] and got[2] in (0, -3)  # (2^32 - 1 entries of 16 bytes twice: fits this device or is an allocation error)
