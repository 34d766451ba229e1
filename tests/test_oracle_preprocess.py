"""Oracle known-answer tests + invariants.  The reference has no golden vectors, so
the oracle is pinned against hand-derived answers and an independent numpy mosaic model."""
import numpy as np
import pytest

import _model as M
import _oracle as O


def run_planar(src, lod_count, T, b, fmt, atlas_size=128, **kw):
    a = O.OracleAtlas(lod_count, atlas_size, False, [(T, b, 1, fmt)])
    a.clear_attachment(0).preprocess_tile(0, src, (0, lod_count), **kw).run()
    return a


def test_task_queue_shape_matches_reference_counts():
    # preprocess_planar.rs: lod_count 4 -> 64 split + 21 downsample + 85 stitch + 85 save
    src = np.full((64, 64), 7, np.uint16)
    a = O.OracleAtlas(4, 1024, False, [(16, 2, 1, O.FORMAT_R16)])
    a.preprocess_tile(0, src, (0, 4))
    n, counts = a.task_counts()
    assert counts == {"split": 64, "downsample": 21, "stitch": 85, "save": 85, "barrier": 3 + 1 + 4}
    tiles = a.tiles()
    # allocation order: finest lod first, x outer / y inner (preprocessor.rs:58-66, 247-268)
    assert tiles[0] == ((0, 3, 0, 0), 0)
    assert tiles[1] == ((0, 3, 0, 1), 1)
    assert tiles[8] == ((0, 3, 1, 0), 8)
    assert tiles[64] == ((0, 2, 0, 0), 64)
    assert tiles[84] == ((0, 0, 0, 0), 84)
    assert len(tiles) == 85


def test_constant_source_gives_constant_tiles():
    src = np.full((100, 100), 12345, np.uint16)
    a = run_planar(src, 3, 16, 2, O.FORMAT_R16)
    for coord, idx in a.tiles():
        assert np.all(a.tile(0, idx) == 12345), coord


def test_split_known_answer_half_texel_average():
    # W == 2^lod * c makes q = g - 0.5: every centre pixel is the exact average of texels g-1 and g
    # (clamped at 0).  Even ramp values make that average an integer, robust against f32 rounding.
    T, b, lod_count = 10, 1, 2
    c = T - 2 * b
    W = (1 << (lod_count - 1)) * c  # 16
    xs = np.arange(W)
    src = (1000 + 2 * xs[None, :] + 100 * xs[:, None]).astype(np.uint16)
    a = run_planar(src, lod_count, T, b, O.FORMAT_R16)
    for gx_tile in range(2):
        for gy_tile in range(2):
            t = a.tile(0, a.get_tile((0, 1, gx_tile, gy_tile)))
            for py in range(b, b + c):
                for px in range(b, b + c):
                    gx, gy = gx_tile * c + px - b, gy_tile * c + py - b
                    x0, x1 = max(gx - 1, 0), gx
                    y0, y1 = max(gy - 1, 0), gy
                    exp = (int(src[y0, x0]) + int(src[y0, x1]) + int(src[y1, x0]) + int(src[y1, x1])) // 4
                    assert t[py, px] == exp, (gx, gy)


def test_downsample_known_answer_and_nodata_rule():
    # finest mosaic = source 2x2-constant blocks -> parent pixel = that constant; zeros are excluded
    T, b = 8, 2
    c = T - 2 * b  # 4
    lod_count = 2
    W = 2 * c  # 8: q = g - 0.5
    # source constant in 4x4 blocks so the half-texel shift stays inside a block for odd g
    src = np.zeros((W, W), np.uint16)
    src[:, :] = 30000
    src[0:4, 0:4] = 0  # a no-data quadrant
    a = run_planar(src, lod_count, T, b, O.FORMAT_R16)
    fine = a.tile(0, a.get_tile((0, 1, 0, 0)))
    # pixels whose footprint touches a zero texel keep the previous (zero) value
    assert fine[b, b] == 0 and fine[b + 3, b + 3] == 0
    root = a.tile(0, a.get_tile((0, 0, 0, 0)))
    # root centre pixel (2,2) <- child (1,1) pixels: all valid 30000
    assert root[b + 2, b + 2] == 30000
    # root pixel (0,0) <- child (0,0) centre px (0..1,0..1): all invalid -> 0 (0/0 defined as 0)
    assert root[b, b] == 0
    # mixed block: average over the valid ones only
    m = M.build_pyramid(src, lod_count, c)
    assert np.array_equal(root[b:b + c, b:b + c], m[0])


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
@pytest.mark.parametrize("T,b,lod_count,W", [(16, 2, 3, 53), (12, 1, 2, 40), (20, 4, 3, 97)])
def test_oracle_equals_mosaic_model(fmt, T, b, lod_count, W):
    rng = np.random.default_rng(1234 + T + W)
    if fmt == O.FORMAT_R16:
        src = rng.integers(1, 65536, size=(W + 3, W), dtype=np.uint16)
        src[rng.random(src.shape) < 0.03] = 0  # no-data holes
    else:
        src = rng.integers(0, 256, size=(W + 3, W, 4), dtype=np.uint8)
        src[..., 0][rng.random(src.shape[:2]) < 0.03] = 0
    c = T - 2 * b
    a = run_planar(src, lod_count, T, b, fmt, atlas_size=64)
    pyr = M.build_pyramid(src, lod_count, c)
    for (side, lod, x, y), idx in a.tiles():
        exp = M.planar_tile_from_mosaic(pyr[lod], lod, x, y, T, b)
        assert np.array_equal(a.tile(0, idx), exp), (lod, x, y)


def test_stitch_border_rules_planar():
    T, b, lod_count = 12, 2, 2
    c = T - 2 * b
    rng = np.random.default_rng(5)
    src = rng.integers(1, 65536, size=(64, 64), dtype=np.uint16)
    a = run_planar(src, lod_count, T, b, O.FORMAT_R16)
    t00 = a.tile(0, a.get_tile((0, 1, 0, 0)))
    t10 = a.tile(0, a.get_tile((0, 1, 1, 0)))
    t01 = a.tile(0, a.get_tile((0, 1, 0, 1)))
    t11 = a.tile(0, a.get_tile((0, 1, 1, 1)))
    o = b + c
    # right apron of (0,0) = first centre columns of (1,0); bottom apron = first centre rows of (0,1)
    assert np.array_equal(t00[b:o, o:], t10[b:o, b:2 * b])
    assert np.array_equal(t00[o:, b:o], t01[b:2 * b, b:o])
    assert np.array_equal(t00[o:, o:], t11[b:2 * b, b:2 * b])
    # missing neighbours: clamp into the own centre; the corner region of an edge tile clamps BOTH axes
    assert np.all(t00[:b, b:o] == t00[b, b:o][None, :])
    assert np.all(t00[b:o, :b] == t00[b:o, b][:, None])
    assert np.all(t00[:b, :b] == t00[b, b])
    assert np.all(t00[:b, o:] == t00[b, o - 1])  # top-right corner: (1,-1) missing -> own (b, o-1)
    assert np.all(t10[:b, :b] == t10[b, b])  # although the left neighbour exists


def test_dataset_subrect_and_partial_coverage():
    # dataset covering x in [0.25, 0.75): only the overlapping tiles exist; split clamps to edge
    T, b, lod_count = 12, 2, 3
    rng = np.random.default_rng(9)
    src = rng.integers(1, 65536, size=(40, 40), dtype=np.uint16)
    a = O.OracleAtlas(lod_count, 64, False, [(T, b, 1, O.FORMAT_R16)])
    a.preprocess_tile(0, src, (0, lod_count), top_left=(0.25, 0.0), bottom_right=(0.75, 0.5)).run()
    coords = {c for c, _ in a.tiles()}
    assert (0, 2, 1, 0) in coords and (0, 2, 2, 1) in coords and (0, 2, 0, 0) not in coords
    assert (0, 2, 3, 0) not in coords and (0, 2, 1, 2) not in coords
    assert (0, 0, 0, 0) in coords
    c = T - 2 * b
    m = M.split_mosaic(src, 2, c, top_left=(0.25, 0.0), bottom_right=(0.75, 0.5))
    t = a.tile(0, a.get_tile((0, 2, 1, 0)))
    assert np.array_equal(t[b:b + c, b:b + c], m[0:c, c:2 * c])


def test_second_dataset_keeps_previous_where_nodata():
    # split.wgsl:37-42: invalid pixels keep the atlas's previous texel
    T, b = 12, 2
    c = T - 2 * b
    base = np.full((32, 32), 20000, np.uint16)
    over = np.full((32, 32), 40000, np.uint16)
    over[:, :16] = 0
    a = O.OracleAtlas(1, 8, False, [(T, b, 1, O.FORMAT_R16)])
    a.preprocess_tile(0, base, (0, 1)).run()
    a.preprocess_tile(0, over, (0, 1)).run()
    t = a.tile(0, 0)[b:b + c, b:b + c]
    assert np.all(t[:, : c // 2 - 1] == 20000)
    assert np.all(t[:, c // 2 + 1:] == 40000)


def test_markstein_unorm_division_is_exact_on_cpu():
    # the fused kernels evaluate t/65535 as q0=t*r, e=fma(-q0,65535,t), q=fma(e,r,q0); check all inputs
    t = np.arange(65536, dtype=np.float32)
    r = np.float32(1.0) / np.float32(65535.0)
    q0 = (t * r).astype(np.float32)
    e = (np.float64(t) - np.float64(q0) * 65535.0).astype(np.float32)  # exact: the true fma result is representable
    q = (np.float64(q0) + np.float64(e) * np.float64(r)).astype(np.float32)
    assert np.array_equal(q, (t / np.float32(65535.0)).astype(np.float32))


def test_sample_tile_known_answers():
    """AtlasAttachment::sample + AttachmentData::sample (tile_atlas.rs:249-258, mod.rs:220-263): hand-computed cases."""
    T, b = 8, 1
    tile = np.zeros((T, T), dtype=np.uint16)
    tile[:, :] = (np.arange(T)[None, :] * 8191).astype(np.uint16)  # a ramp in x: texel x holds x * 8191 (65528 at x = 8)
    # uv = atlas_uv * (6/8) + 1/8; pixel = uv * 8 - 0.5.  atlas_uv.x = 0.5 -> uv 0.5 -> pixel 3.5: half-way texels 3 and 4
    s = O.sample_tile(O.FORMAT_R16, b, tile, (0.5, 0.5))
    a, c = np.float32(3 * 8191) / np.float32(65535), np.float32(4 * 8191) / np.float32(65535)
    assert s[0] == np.float32(a + (c - a) * np.float32(0.5)) and s[1] == s[2] == s[3] == 0
    # atlas_uv = 0 -> pixel 0.5: half-way between the apron texel 0 and the first centre texel 1
    s = O.sample_tile(O.FORMAT_R16, b, tile, (0.0, 0.25))
    assert s[0] == np.float32(np.float32(8191) / np.float32(65535) * np.float32(0.5))
    # RGBA8: constant colour comes back exactly, all four channels
    rgba = np.zeros((T, T, 4), dtype=np.uint8)
    rgba[...] = (255, 128, 1, 77)
    s = O.sample_tile(O.FORMAT_RGBA8, b, rgba, (0.3, 0.9))
    assert np.array_equal(s, (np.array([255, 128, 1, 77], dtype=np.float32) / np.float32(255)).astype(np.float32))


def test_block_schedule_and_native_build_give_the_same_bytes(tmp_path):
    """bench.py's CPU-baseline leg runs the oracle with another schedule (units = (task, 32-row block), in-place stores:
    orc_run_blocks) and another build (-O3 -march=native, still -ffp-contract=off): neither may change one byte"""
    import ctypes as C
    import os
    import subprocess

    src = np.random.default_rng(5).integers(0, 65536, size=(300, 280), dtype=np.uint16)
    src[src < 3000] = 0
    ref = run_planar(src, 4, 32, 2, O.FORMAT_R16)
    native = os.path.join(O.ORACLE_DIR, "libbt_oracle_native.so")
    subprocess.check_call(["make", "-C", O.ORACLE_DIR, "-s", "-B", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for path in (O._LIB_PATH, native):
        L = C.CDLL(path)
        vp, u32 = C.c_void_p, C.c_uint32
        L.orc_atlas_new.argtypes = [u32, u32, C.c_int, u32, C.POINTER(O.AttachmentConfig)]
        L.orc_atlas_new.restype = vp
        L.orc_preprocess_tile.argtypes = [vp, C.POINTER(O.Dataset), vp, u32, u32]
        L.orc_run_blocks.argtypes = [vp, C.c_int, u32, C.POINTER(C.c_double), C.POINTER(u32), u32, C.POINTER(u32)]
        L.orc_atlas_touch.argtypes = [vp, C.c_int]
        L.orc_tile_data.argtypes = [vp, u32, u32]
        L.orc_tile_data.restype = vp
        L.orc_atlas_free.argtypes = [vp]
        for threads, rows in ((1, 32), (4, 7), (3, 64)):
            cfg = (O.AttachmentConfig * 1)(O.AttachmentConfig(32, 2, 1, O.FORMAT_R16))
            h = L.orc_atlas_new(4, 128, 0, 1, cfg)
            d = O.Dataset(0, 0, (C.c_float * 2)(0.0, 0.0), (C.c_float * 2)(1.0, 1.0), 0, 4)
            assert L.orc_preprocess_tile(h, C.byref(d), src.ctypes.data, src.shape[1], src.shape[0]) == 0
            L.orc_atlas_touch(h, threads)
            secs, tasks, n = (C.c_double * 16)(), (u32 * 16)(), u32()
            assert L.orc_run_blocks(h, threads, rows, secs, tasks, 16, C.byref(n)) == 0
            assert [tasks[i] for i in range(n.value)] == [64, 16, 4, 1, 1, 4, 16, 64]  # split, 3 x downsample, 4 x stitch
            for coord, idx in ref.tiles():
                got = np.frombuffer((C.c_uint8 * 2048).from_address(L.orc_tile_data(h, 0, idx)), dtype=np.uint16).reshape(32, 32)
                assert np.array_equal(got, ref.tile(0, idx)), (path, threads, rows, coord)
            L.orc_atlas_free(h)
