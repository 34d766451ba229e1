"""Parity of the HIP preprocess path (through the C ABI) against the CPU oracle: bit-exact tiles."""
import os

import numpy as np
import pytest

import _cases as K
import _model as M
import _oracle as O
import bevy_terrain_amd as bt

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def device():
    d = bt.Device(0)
    yield d


@pytest.mark.parametrize("generic", [True, False])
@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
@pytest.mark.parametrize("T,b,lod_count,W,H", [(16, 2, 3, 53, 56), (12, 1, 2, 40, 40), (64, 2, 3, 300, 257), (20, 4, 3, 97, 97)])
def test_planar_random_with_holes(device, generic, fmt, T, b, lod_count, W, H):
    src = K.random_raster(fmt, H, W, seed=T * 1000 + W, holes=0.03)
    atlas, _ = K.product_planar(device, src, lod_count, T, b, fmt, generic=generic)
    oracle = K.oracle_planar(src, lod_count, T, b, fmt)
    assert K.assert_atlas_equal(atlas, oracle) == sum(4 ** l for l in range(lod_count))


@pytest.mark.parametrize("generic", [True, False])
def test_planar_reference_tile_shape_512(device, generic):
    # the reference's tile shape (T=512, b=2) on a 2k raster, lod_count 3 -> 21 tiles
    src = K.smooth_raster(2048, 2048, seed=1234, device=device)
    src[100:140, 900:1000] = 0
    atlas, pre = K.product_planar(device, src, 3, 512, 2, O.FORMAT_R16, generic=generic)
    oracle = K.oracle_planar(src, 3, 512, 2, O.FORMAT_R16)
    assert K.assert_atlas_equal(atlas, oracle) == 21
    st = pre.stats()
    assert st["tiles"] == 21 and st["algorithmic_bytes"] == 2048 * 2048 * 2 + 21 * 512 * 512 * 2


def test_config2_planar_4k_height_and_albedo(device):
    """BASELINE config 2: 4096^2 height (R16) + albedo (Rgba8), lod_count 4, 85 tiles each, bit-compared with the
    reference's own WGSL executed on the CPU (oracle/_ref)."""
    height = K.smooth_raster(4096, 4096, seed=1234, device=device)
    rng = np.random.default_rng(1235)
    albedo = rng.integers(1, 256, size=(4096, 4096, 4), dtype=np.uint8)
    albedo[..., 3] = 255
    cfg = bt.TerrainConfig(lod_count=4, path="terrains/planar", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=512, border_size=2, format=bt.AttachmentFormat.Rgba8))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("terrains/planar/source/height.png", height).insert("terrains/planar/source/albedo.png", albedo)
    pre = (bt.Preprocessor.new().clear_attachment(0, atlas).clear_attachment(1, atlas)
           .preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="terrains/planar/source/height.png", lod_range=range(0, 4)), server, atlas)
           .preprocess_tile(bt.PreprocessDataset(attachment_index=1, path="terrains/planar/source/albedo.png", lod_range=range(0, 4)), server, atlas))
    assert pre.task_counts() == {"split": 128, "stitch": 170, "downsample": 42, "save": 170, "barrier": 16}
    pre.run(atlas)
    oracle = K.reference_kernels(O.OracleAtlas(4, 1024, False, [(512, 2, 1, O.FORMAT_R16), (512, 2, 1, O.FORMAT_RGBA8)]))  # executed WGSL
    oracle.clear_attachment(0).clear_attachment(1)
    oracle.preprocess_tile(0, height, (0, 4)).preprocess_tile(1, albedo, (0, 4)).run(8)
    assert K.assert_atlas_equal(atlas, oracle, 0) == 85
    assert K.assert_atlas_equal(atlas, oracle, 1) == 85


@pytest.mark.parametrize("generic", [True, False])
def test_dataset_subrect(device, generic):
    src = K.random_raster(O.FORMAT_R16, 80, 90, seed=9)
    ds = dict(top_left=(0.25, 0.0), bottom_right=(0.75, 0.5))
    atlas, _ = K.product_planar(device, src, 3, 16, 2, O.FORMAT_R16, generic=generic, **ds)
    oracle = K.oracle_planar(src, 3, 16, 2, O.FORMAT_R16, **ds)
    assert K.assert_atlas_equal(atlas, oracle) > 0


@pytest.mark.parametrize("generic", [True, False])
def test_overlay_keeps_previous_where_nodata(device, generic):
    T, b = 32, 2
    base = K.random_raster(O.FORMAT_R16, 100, 100, seed=1)
    over = K.random_raster(O.FORMAT_R16, 64, 64, seed=2, holes=0.3)
    cfg = bt.TerrainConfig(lod_count=2, atlas_size=16, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("base", base).insert("over", over)
    pre = bt.Preprocessor.new()
    pre.preprocess_tile(bt.PreprocessDataset(path="base", lod_range=range(0, 2)), server, atlas).run(atlas, generic=generic)
    pre.preprocess_tile(bt.PreprocessDataset(path="over", lod_range=range(0, 2)), server, atlas).run(atlas, generic=generic)
    oracle = O.OracleAtlas(2, 16, False, [(T, b, 1, O.FORMAT_R16)])
    oracle.preprocess_tile(0, base, (0, 2)).run()
    oracle.preprocess_tile(0, over, (0, 2)).run()
    assert K.assert_atlas_equal(atlas, oracle) == 5


@pytest.mark.parametrize("generic", [True, False])
@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_spherical_cube_faces(device, generic, fmt):
    T, b, lod_count, W = 32, 2, 3, 100
    faces = [K.random_raster(fmt, W, W, seed=70 + s, holes=0.02) for s in range(6)]
    cfg = bt.TerrainConfig(lod_count=lod_count, atlas_size=256, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=K.FMT[fmt]))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    paths = [f"face{s}" for s in range(6)]
    for p, f in zip(paths, faces):
        server.insert(p, f)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lod_count)), server, atlas)
    pre.run(atlas, generic=generic)
    oracle = O.OracleAtlas(lod_count, 256, True, [(T, b, 1, fmt)])
    oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lod_count)).run(8)
    assert K.assert_atlas_equal(atlas, oracle) == 6 * 21


def test_atlas_out_of_indices_is_an_error_not_a_panic(device):
    src = K.random_raster(O.FORMAT_R16, 64, 64, seed=3)
    with pytest.raises(bt._ffi.BtError) as e:
        K.product_planar(device, src, 3, 16, 2, O.FORMAT_R16, atlas_size=20)
    assert e.value.status == -2


def test_unsupported_formats_are_rejected(device):
    cfg = bt.TerrainConfig(lod_count=1, atlas_size=4, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="n", texture_size=16, border_size=1, format=bt.AttachmentFormat.Rg16))
    atlas = bt.TileAtlas.new(cfg, device)
    with pytest.raises(Exception):
        bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="x"), bt.AssetServer().insert("x", np.zeros((8, 8), np.uint16)), atlas)


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_generate_mipmaps_exact(device, fmt):
    T = 64
    level0 = K.random_raster(fmt, T, T, seed=11, holes=0.2 if fmt == O.FORMAT_R16 else 0.0)
    ours = bt.generate_mipmaps(device, K.FMT[fmt], level0, 5)
    assert np.array_equal(ours, O.generate_mipmaps(fmt, level0, 5))


def test_atlas_mip_chain_and_files(device, tmp_path):
    src = K.random_raster(O.FORMAT_R16, 200, 200, seed=21, holes=0.05)
    atlas, pre = K.product_planar(device, src, 3, 32, 2, O.FORMAT_R16, mips=4)
    oracle = K.oracle_planar(src, 3, 32, 2, O.FORMAT_R16)
    atlas.generate_mipmaps(0, 0, 21)
    for (coord, idx) in oracle.tiles()[::5]:
        chain = O.generate_mipmaps(O.FORMAT_R16, oracle.tile(0, idx), 4)
        off = 32 * 32
        for mip in (1, 2, 3):
            s = 32 >> mip
            assert np.array_equal(atlas.download_mip(0, mip, idx).ravel(), chain[off:off + s * s]), (coord, mip)
            off += s * s
    # files: "{root}/{path}/data/{name}/{side}_{lod}_{x}_{y}.bin" + config.tc, byte-identical to the oracle's
    root = str(tmp_path / "assets")
    pre.save(atlas, root)
    d = os.path.join(root, "terrains/test/data/att")
    odir = str(tmp_path / "oracle")
    os.makedirs(odir)
    oracle.save_attachment(0, odir)
    names = sorted(os.listdir(d))
    assert names == sorted(os.listdir(odir)) and len(names) == 21 and "0_2_3_1.bin" in names
    for n in names:
        assert open(os.path.join(d, n), "rb").read() == open(os.path.join(odir, n), "rb").read()
    tc = open(os.path.join(root, "terrains/test/config.tc"), "rb").read()
    assert sorted((t.side, t.lod, t.x, t.y) for t in bt.tc_decode(tc)) == sorted(c for c, _ in oracle.tiles())
    assert set(O.tc_decode(tc)) == {c for c, _ in oracle.tiles()}


def test_synth_fbm_matches_integer_model(device):
    w, h = 300, 200
    ptr = device.synth_fbm_r16(w, h, seed=42, x0=1000, y0=77, base_cell=64)
    ours = device.download(ptr, (h, w), np.uint16)
    device.free(ptr)
    exp = M.fbm_u16(w, h, 42, x0=1000, y0=77, base_cell=64)
    assert np.array_equal(ours, exp)
    assert ours.min() >= 1


def test_fast_unorm_conversion_is_exact_on_device(device):
    import ctypes as C
    failures = C.c_uint32(123)
    bt._ffi.check(bt._ffi.lib().bt_selftest(device._h, C.byref(failures)))
    assert failures.value == 0


@pytest.mark.parametrize("generic", [True, False])
def test_saturated_and_flat_rasters(device, generic):
    # blends of all-65535 texels round to 1 + ulp before the clamp: the fused fast loop drops the clamp, so pin it
    src = np.full((1100, 1100), 65535, np.uint16)
    src[300:500, :] = 65534
    src[:, 700:] = 1
    src[900:, 900:] = K.random_raster(O.FORMAT_R16, 200, 200, seed=77)
    atlas, _ = K.product_planar(device, src, 4, 512, 2, O.FORMAT_R16, atlas_size=128, generic=generic)
    oracle = K.oracle_planar(src, 4, 512, 2, O.FORMAT_R16, atlas_size=128)
    assert K.assert_atlas_equal(atlas, oracle) == 85


@pytest.mark.parametrize("lane", range(8))
def test_single_nodata_texel_is_seen_at_every_position_of_a_staging_load(device, lane):
    # one isolated no-data texel: the fast path must hand exactly its chunk to the validity-aware variant,
    # whichever of the 8 texels of a 16-byte staging load it is
    src = K.random_raster(O.FORMAT_R16, 600, 640, seed=5)
    src[301, 320 + lane] = 0
    atlas, pre = K.product_planar(device, src, 3, 128, 2, O.FORMAT_R16)
    assert pre.stats()["fused_jobs"] == 1
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, 3, 128, 2, O.FORMAT_R16)) == 21


def test_config5_cube_8k_faces_full_size(device):
    """BASELINE config 5 at full size: 6 cube faces of 8192^2 (R16), lod_count 5, T = 512 -> 2046 tiles through the
    fused path (cube seams re-stitched by the batched kernel), every tile compared with the reference's own WGSL executed on
    the CPU (oracle/_ref)."""
    W, lods = 8192, 5
    faces = [K.smooth_raster(W, W, seed=7 + s, device=device) for s in range(6)]
    for s in range(6):
        faces[s][1000 + 37 * s:1100 + 37 * s, 5000:5300] = 0  # a no-data patch on every face
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    paths = [f"face{s}" for s in range(6)]
    for p, f in zip(paths, faces):
        server.insert(p, f)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
    pre.run(atlas)
    assert pre.stats()["fused_jobs"] >= 1 and pre.stats()["tiles"] == 2046
    oracle = K.reference_kernels(O.OracleAtlas(lods, 2048, True, [(512, 2, 1, O.FORMAT_R16)]))  # the reference's WGSL, executed
    oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
    assert [((c.side, c.lod, c.x, c.y), i) for c, i in atlas.tiles()] == oracle.tiles()
    for first in range(0, 2046, 128):
        count = min(128, 2046 - first)
        data = atlas.download_tiles(0, first, count)
        for k in range(count):
            assert np.array_equal(data[k], oracle.tile(0, first + k)), (first + k, oracle.tiles()[first + k])


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_tile_load_path_round_trip(device, tmp_path, fmt):
    """preprocess -> save -> (fresh atlas) load_tile_config + load_tiles: level 0 comes back byte
    for byte, the mip levels equal the reference's CPU generate_mipmaps of every tile, a missing file is an IO error."""
    T, lods = 32, 3
    src = K.random_raster(fmt, 200, 200, seed=31, holes=0.05)
    atlas, pre = K.product_planar(device, src, lods, T, 2, fmt, mips=4)
    root = str(tmp_path / "assets")
    pre.save(atlas, root)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=64, path="terrains/test", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=2, format=K.FMT[fmt], mip_level_count=4))
    fresh = bt.TileAtlas.new(cfg, device)
    fresh.load_tile_config(root)
    fresh.load_tiles(0, root)
    loaded = {(c.side, c.lod, c.x, c.y): i for c, i in fresh.tiles()}
    original = {(c.side, c.lod, c.x, c.y): i for c, i in atlas.tiles()}
    assert set(loaded) == set(original) and len(loaded) == 21
    ch = 1 if fmt == O.FORMAT_R16 else 4
    for coord, idx in loaded.items():
        level0 = fresh.download_tile(0, idx)
        assert np.array_equal(level0, atlas.download_tile(0, original[coord])), coord
        chain = np.asarray(O.generate_mipmaps(fmt, level0, 4))
        off = T * T * ch
        for mip in (1, 2, 3):
            s = T >> mip
            assert np.array_equal(fresh.download_mip(0, mip, idx).ravel(), chain[off:off + s * s * ch]), (coord, mip)
            off += s * s * ch
    # explicit coordinate list + error path
    os.remove(os.path.join(root, "terrains/test/data/att", "0_2_3_1.bin"))
    again = bt.TileAtlas.new(cfg, device)
    again.load_tiles(0, root, [bt.TileCoordinate(0, 0, 0, 0), bt.TileCoordinate(0, 1, 1, 0)])
    assert len(again.tiles()) == 2
    with pytest.raises(bt._ffi.BtError) as e:
        again.load_tiles(0, root, [bt.TileCoordinate(0, 2, 3, 1)])
    assert e.value.status == -5 or "not found" in str(e.value)
    # a file of the wrong length (cut short, or longer than a tile) is an error too, not a partly loaded tile
    for name, change in (("0_2_0_0.bin", lambda d: d[:-7]), ("0_2_1_1.bin", lambda d: d + b"\0")):
        path = os.path.join(root, "terrains/test/data/att", name)
        data = open(path, "rb").read()
        open(path, "wb").write(change(data))
        side, lod, x, y = (int(v) for v in name[:-4].split("_"))
        with pytest.raises(bt._ffi.BtError) as e:
            bt.TileAtlas.new(cfg, device).load_tiles(0, root, [bt.TileCoordinate(side, lod, x, y)])
        assert "does not hold" in str(e.value), str(e.value)


@pytest.mark.parametrize("T,b,W", [(512, 4, 2100), (512, 8, 1900), (256, 2, 1100), (384, 6, 1500)])
def test_fused_path_other_borders_and_sizes(device, T, b, W):
    # wider aprons / other texture sizes through the fused kernels (runtime-shape template instance for T != 512 or b != 2)
    src = K.smooth_raster(W, W + 64, seed=T + b, device=device)
    src[W // 2:W // 2 + 9, W // 3:W // 3 + 40] = 0
    atlas, pre = K.product_planar(device, src, 3, T, b, O.FORMAT_R16)
    assert pre.stats()["fused_jobs"] == 1
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, 3, T, b, O.FORMAT_R16)) == 21


@pytest.mark.parametrize("W,H", [(200, 170), (2500, 2300), (4100, 600)])
def test_fused_path_resample_ratios(device, W, H):
    # mosaic of 4 x 124 = 496 pixels per side: upsampling (0.4x), strong downsampling (5x: the source window no longer
    # fits the LDS budget, the kernel variant that reads the raster directly takes over) and a very anisotropic raster
    src = K.random_raster(O.FORMAT_R16, H, W, seed=W, holes=0.002)
    atlas, pre = K.product_planar(device, src, 3, 128, 2, O.FORMAT_R16)
    assert pre.stats()["fused_jobs"] == 1
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, 3, 128, 2, O.FORMAT_R16)) == 21


@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("pad_texels", [8, 3])
def test_padded_row_pitch_device_raster(device, generic, pad_texels):
    # a device raster whose rows are padded (row_pitch > width * 2): 16-byte aligned pitch takes the wide staging
    # loads, an odd pitch the texel-by-texel staging
    h, w = 300, 520
    src = K.random_raster(O.FORMAT_R16, h, w, seed=77, holes=0.01)
    padded = np.full((h, w + pad_texels), 0xABCD, dtype=np.uint16)
    padded[:, :w] = src
    ptr = device.upload(padded)
    cfg = bt.TerrainConfig(lod_count=3, atlas_size=64, path="terrains/test", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=128, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("src", (ptr, w, h, (w + pad_texels) * 2))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
        bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, 3)), server, atlas)
    pre.run(atlas, generic=generic)
    device.free(ptr)
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, 3, 128, 2, O.FORMAT_R16)) == 21


def test_degenerate_inputs(device):
    cfg = bt.TerrainConfig(lod_count=3, atlas_size=32, path="terrains/test", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=16, border_size=2, format=bt.AttachmentFormat.R16))
    src = K.random_raster(O.FORMAT_R16, 40, 40, seed=5)
    # an empty LOD range (the reference underflows `lod_range.end - 1`, preprocessor.rs:300-312) is an argument error,
    # and nothing is queued; running an empty queue is a no-op
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas)
    with pytest.raises(bt._ffi.BtError) as e:
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="s", lod_range=range(0, 0)), bt.AssetServer().insert("s", src), atlas)
    assert e.value.status == -1
    assert sum(pre.task_counts().values()) == 0
    pre.run(atlas)
    assert atlas.tiles() == []
    # a 1 x 1 raster: every pixel of every tile is that texel (clamp-to-edge), through both paths
    one = np.array([[12345]], dtype=np.uint16)
    for generic in (False, True):
        a, _ = K.product_planar(device, one, 2, 16, 2, O.FORMAT_R16, generic=generic)
        assert K.assert_atlas_equal(a, K.oracle_planar(one, 2, 16, 2, O.FORMAT_R16)) == 5
        assert int(a.download_tile(0, 0)[8, 8]) == 12345


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_atlas_sample_matches_the_cpu_sampling(device, fmt):
    """SURVEY §8f row 4, the atlas half: TileAtlas::sample_attachment as a batched query on the tiles in HBM ==
    the reference's CPU sampling (oracle restatement) bit for bit, INVALID lookups give zero."""
    T, b = 32, 2
    src = K.random_raster(fmt, 200, 200, seed=41)
    atlas, _ = K.product_planar(device, src, 3, T, b, fmt)
    rng = np.random.default_rng(9)
    n = 2000
    idx = rng.integers(0, 21, size=n).astype(np.uint32)
    uv = rng.random((n, 2), dtype=np.float32)
    uv[:8] = [[0, 0], [1, 1], [0, 1], [1, 0], [0.5, 0.5], [1e-7, 0.999999], [0.25, 0.75], [0.999, 0.001]]
    idx[-1] = 0xFFFFFFFF
    ours = atlas.sample(0, idx, uv)
    tiles = atlas.download_tiles(0, 0, 21)
    for i in range(n - 1):
        exp = O.sample_tile(fmt, b, tiles[idx[i]], uv[i])
        assert np.array_equal(ours[i], exp), (i, idx[i], uv[i], ours[i], exp)
    assert np.array_equal(ours[-1], np.zeros(4, np.float32))
    if fmt == O.FORMAT_R16:  # sample_height = lerp(min_height, max_height, value.x): heights stay inside the texel range
        assert ours[:-1, 0].min() >= tiles.min() / 65535.0 - 1e-6 and ours[:-1, 0].max() <= tiles.max() / 65535.0 + 1e-6


@pytest.mark.parametrize("lod_count,T,W", [(1, 64, 100), (2, 64, 150), (8, 16, 1600)])
def test_fused_path_shallow_and_deep_pyramids(device, lod_count, T, W):
    # one LOD (no pyramid at all), two (no tail launch: the apron rows of the parent come from the rows-only stitch),
    # eight (fused_main + two tail launches, 21845 tiles)
    src = K.random_raster(O.FORMAT_R16, W, W + 7, seed=lod_count, holes=0.001)
    n = sum(4 ** l for l in range(lod_count))
    atlas, pre = K.product_planar(device, src, lod_count, T, 2, O.FORMAT_R16, atlas_size=n + 3)
    assert pre.stats()["fused_jobs"] == 1
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, lod_count, T, 2, O.FORMAT_R16, atlas_size=n + 3)) == n


def _nodata_mask_16k(seed=43):
    """the 16k variant with a 5 % zero "no-data" mask (seed 43).  5 % of the 37 x 53 texel cells of the
    raster are zeroed (odd cell sizes, so the holes meet every staging-load lane, chunk and tile edge) plus one
    isolated no-data texel per 2^14 texels."""
    rng = np.random.default_rng(seed)
    cells = rng.random((16384 // 37 + 1, 16384 // 53 + 1)) < 0.05
    mask = np.repeat(np.repeat(cells, 37, axis=0), 53, axis=1)[:16384, :16384]
    single = rng.integers(0, 16384, size=(16384, 2))
    mask[single[:, 0], single[:, 1]] = True
    return mask


@pytest.mark.parametrize("masked", [False, True])
def test_config3_16k_single_gpu_all_tiles(device, masked):
    """BASELINE config 3 at N = 1, the workload bench.py times: 16384^2 fBm (seed 42), T = 512, b = 2, lod_count 6 ->
    1365 tiles through the fused plan (1024 workgroups, fused_main / fused_tail), every tile byte-compared
    with the reference's own WGSL executed on the CPU (oracle/_ref); the masked variant (seed 43, 5 % no-data) drives fused_main's no-data redo at full size."""
    size, lods = 16384, 6
    ptr = device.synth_fbm_r16(size, size, 42)
    src = device.download(ptr, (size, size), np.uint16)
    if masked:
        hole = _nodata_mask_16k()
        assert 0.04 < hole.mean() < 0.06
        src[hole] = 0
        device.free(ptr)
        ptr = device.upload(src)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/bench16k",
                           model=bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("synthetic/fbm16k", (ptr, size, size))
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
        bt.PreprocessDataset(attachment_index=0, path="synthetic/fbm16k", lod_range=range(0, lods)), server, atlas)
    pre.run(atlas, keep_queue=True)
    pre.run(atlas)  # a second run of the same queue (the todo lists alternate between runs) must give the same atlas
    device.free(ptr)
    st = pre.stats()
    assert st["fused_jobs"] == 1 and st["tiles"] == 1365 and st["algorithmic_bytes"] == 1252524032
    oracle = K.reference_kernels(O.OracleAtlas(lods, 2048, False, [(512, 2, 1, O.FORMAT_R16)]))  # the reference's WGSL, executed
    oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(O.usable_cores())
    assert [((c.side, c.lod, c.x, c.y), i) for c, i in atlas.tiles()] == oracle.tiles()
    for first in range(0, 1365, 128):
        count = min(128, 1365 - first)
        data = atlas.download_tiles(0, first, count)
        for k in range(count):
            assert np.array_equal(data[k], oracle.tile(0, first + k)), (first + k, oracle.tiles()[first + k])


@pytest.mark.parametrize("second_run", [False, True])
def test_adjacent_datasets_stitch_across_their_seam(device, second_run):
    """Two datasets covering the left and the right half of the terrain (PreprocessDataset::top_left / bottom_right),
    LODs 1..3 only (so no shared ancestor is queued and both quadtrees are complete): the aprons along the seam are the
    OTHER dataset's centres (stitch_and_save_layer records the atlas's neighbours).  The fused path stitches from a
    job's own grid only, so such jobs must fall back to the batched kernels: BT_RUN_AUTO == BT_RUN_GENERIC == oracle.
    second_run: the right half is queued and run after the left half has finished (an earlier run on the same atlas)."""
    T, b = 32, 2
    left = K.random_raster(O.FORMAT_R16, 130, 70, seed=51, holes=0.01)
    right = K.random_raster(O.FORMAT_R16, 130, 66, seed=52, holes=0.01)
    halves = [("left", left, dict(top_left=(0.0, 0.0), bottom_right=(0.5, 1.0))), ("right", right, dict(top_left=(0.5, 0.0), bottom_right=(1.0, 1.0)))]
    results = []
    for generic in (True, False):
        cfg = bt.TerrainConfig(lod_count=3, atlas_size=64, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer().insert("left", left).insert("right", right)
        pre = bt.Preprocessor.new()
        for name, _, rect in halves:
            pre.preprocess_tile(bt.PreprocessDataset(path=name, lod_range=range(1, 3), **rect), server, atlas)
            if second_run:
                pre.run(atlas, generic=generic)
        if not second_run:
            pre.run(atlas, generic=generic)
        results.append(atlas)
    oracle = O.OracleAtlas(3, 64, False, [(T, b, 1, O.FORMAT_R16)])
    for _, src, rect in halves:
        oracle.preprocess_tile(0, src, (1, 3), **rect)
        if second_run:
            oracle.run()
    if not second_run:
        oracle.run()
    for atlas in results:
        assert K.assert_atlas_equal(atlas, oracle) == 4 + 16


def test_dataset_rectangle_is_clamped_to_the_face(device):
    # bottom_right > 1 and a negative top_left: `as_uvec2` saturates, tiles beyond the face do not exist
    src = K.random_raster(O.FORMAT_R16, 64, 64, seed=8)
    ds = dict(top_left=(-0.25, 0.0), bottom_right=(1.5, 1.0))
    for generic in (True, False):
        atlas, _ = K.product_planar(device, src, 3, 16, 2, O.FORMAT_R16, generic=generic, **ds)
        coords = [(c.lod, c.x, c.y) for c, _ in atlas.tiles()]
        assert len(coords) == 21 and all(x < (1 << lod) and y < (1 << lod) for lod, x, y in coords)


def test_raster_pitch_is_validated(device):
    cfg = bt.TerrainConfig(lod_count=1, atlas_size=4, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=16, border_size=2))
    atlas = bt.TileAtlas.new(cfg, device)
    ptr = device.upload(np.ones((8, 8), np.uint16))
    for pitch in (15, 14):  # odd, and shorter than a row
        with pytest.raises(bt._ffi.BtError) as e:
            bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="x"), bt.AssetServer().insert("x", (ptr, 8, 8, pitch)), atlas)
        assert e.value.status == -1
    device.free(ptr)
    # center_size < border_size cannot be stitched in place (the reference would read aprons being written)
    cfg2 = bt.TerrainConfig(lod_count=1, atlas_size=4, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg2.add_attachment(bt.AttachmentConfig(name="h", texture_size=16, border_size=6))
    with pytest.raises(bt._ffi.BtError) as e:
        bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="x"), bt.AssetServer().insert("x", np.ones((8, 8), np.uint16)), bt.TileAtlas.new(cfg2, device))
    assert e.value.status == -5


def test_preprocess_from_png_and_tiff_files(device, tmp_path):
    """examples/preprocess_planar.rs end to end from image FILES: `asset_server.load("terrains/planar/source/height.png")`
    is the library's PNG decoder here (bt_image_load), a TIFF face like examples/preprocess_spherical.rs' as well."""
    from PIL import Image

    root = tmp_path / "assets"
    os.makedirs(root / "terrains/planar/source")
    height = K.smooth_raster(700, 700, seed=12)
    albedo = np.random.default_rng(13).integers(1, 256, size=(700, 700, 3), dtype=np.uint8)
    Image.fromarray(height).save(str(root / "terrains/planar/source/height.png"))
    Image.fromarray(albedo).save(str(root / "terrains/planar/source/albedo.png"))
    Image.fromarray(height).save(str(root / "terrains/planar/source/height.tif"), compression="tiff_lzw")
    cfg = bt.TerrainConfig(lod_count=3, atlas_size=128, path="terrains/planar", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=128, border_size=2, format=bt.AttachmentFormat.R16))
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=128, border_size=2, format=bt.AttachmentFormat.Rgba8))
    cfg.add_attachment(bt.AttachmentConfig(name="height_tif", texture_size=128, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer(str(root))
    pre = bt.Preprocessor.new()
    for i, name in enumerate(("height.png", "albedo.png", "height.tif")):
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=i, path=f"terrains/planar/source/{name}", lod_range=range(0, 3)), server, atlas)
    pre.run(atlas)
    rgba = np.concatenate([albedo, np.full((700, 700, 1), 255, np.uint8)], axis=2)
    oracle = O.OracleAtlas(3, 128, False, [(128, 2, 1, O.FORMAT_R16), (128, 2, 1, O.FORMAT_RGBA8), (128, 2, 1, O.FORMAT_R16)])
    oracle.preprocess_tile(0, height, (0, 3)).preprocess_tile(1, rgba, (0, 3)).preprocess_tile(2, height, (0, 3)).run(8)
    for i in range(3):
        assert K.assert_atlas_equal(atlas, oracle, i) == 21


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_two_contexts_interleaved_on_one_gpu_stay_exact(device, fmt):
    """Independent jobs in flight: two contexts (a HIP stream each) with their own atlas and source take turns without
    any synchronisation in between — the kernels of one job run beside the other's (what bench.py --pipeline 2 does).
    Sources with holes, so the todo path and its alternating lists are exercised too."""
    T, b, lods = 256, 2, 4
    srcs = [K.random_raster(fmt, 1000, 1100, 900 + k, holes=0.03) for k in range(2)]
    devices = [device, bt.Device(device.index)]
    jobs = []
    for d, src in zip(devices, srcs):
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=128, path="terrains/test", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
        atlas = bt.TileAtlas.new(cfg, d)
        server = bt.AssetServer().insert("src", src)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
            bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods)), server, atlas)
        jobs.append((atlas, pre, server))
    for k in (0, 1, 0, 1, 1, 0, 0, 1):
        atlas, pre, _ = jobs[k]
        pre.run(atlas, keep_queue=True, sync=False)
    assert jobs[0][1].stats()["fused_jobs"] == 1
    for (atlas, _, _), src in zip(jobs, srcs):
        assert K.assert_atlas_equal(atlas, K.oracle_planar(src, lods, T, b, fmt)) == 85


@pytest.mark.parametrize("cube", [False, True])
@pytest.mark.parametrize("T,b", [(16, 4), (24, 4), (20, 2)])
def test_fused_path_narrow_tiles_wide_borders(device, cube, T, b):
    """c / 4 < b: the b-wide apron strip of a grand-parent tile spans the shares of more than one finest tile (found by
    the random sweep: the x pushes of fused_main assumed the strip lies inside the first / last finest tile's share)."""
    lods, fmt = 4, O.FORMAT_R16
    if cube:
        faces = [K.random_raster(fmt, 21, 21, 49 + s, holes=0.3) for s in range(6)]
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=600, path="terrains/narrow")
        cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer()
        for s in range(6):
            server.insert(f"f{s}", faces[s])
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
            bt.SphericalDataset(attachment_index=0, paths=[f"f{s}" for s in range(6)], lod_range=range(0, lods)), server, atlas)
        pre.run(atlas)
        assert pre.stats()["fused_jobs"] == 1
        oracle = O.OracleAtlas(lods, 600, True, [(T, b, 1, fmt)])
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(8)
        assert K.assert_atlas_equal(atlas, oracle) == 6 * 85
    else:
        for holes in (0.0, 0.3):
            src = K.random_raster(fmt, 70, 61, 5, holes=holes)
            atlas, pre = K.product_planar(device, src, lods, T, b, fmt)
            assert pre.stats()["fused_jobs"] == 1
            assert K.assert_atlas_equal(atlas, K.oracle_planar(src, lods, T, b, fmt)) == 85


@pytest.mark.parametrize("T,b,lod_count,W,holes", [(64, 2, 6, 1930, 0.0), (64, 2, 6, 1900, 0.01), (100, 2, 5, 1560, 0.002), (36, 2, 6, 1000, 0.0), (260, 4, 4, 2100, 0.003)])
def test_direct_rgba8_several_row_blocks_per_workgroup(device, T, b, lod_count, W, holes):
    # fused_direct with many small tiles: the planner gives a workgroup several 8-row blocks (8 blocks of a 64^2 tile =
    # the whole tile, 3 of the 12 blocks of a 100^2 tile, a ragged last block for c = 32, apron columns outside the sweeps for c + 2b > 256 ...), with and without no-data
    # texels (the wave-uniform fast path and the per-pixel path side by side), the parents' aprons from the tail launch
    src = K.random_raster(O.FORMAT_RGBA8, W + 30, W, seed=T * lod_count, holes=holes)
    tiles = sum(4 ** l for l in range(lod_count))
    atlas, pre = K.product_planar(device, src, lod_count, T, b, O.FORMAT_RGBA8, atlas_size=2048)
    assert pre.stats()["fused_jobs"] == 1
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, lod_count, T, b, O.FORMAT_RGBA8, atlas_size=2048)) == tiles


@pytest.mark.parametrize("holes", [False, True])
def test_streamed_run_writes_the_same_files(device, tmp_path, holes):
    """bt_preprocessor_run_streamed (upload in bands of tile rows || kernels || download + file writes) against run() + save()
    of the same job, file by file, and against the oracle's atlas: 8192^2 R16, lod_count 5 -> 256 finest tiles in 4 bands of 4
    tile rows; with no-data patches (fused_main's redo runs band by band) and without."""
    size, lods = 8192, 5
    src = K.smooth_raster(size, size, seed=91, device=device)
    if holes:
        src[700:900, 3000:3400] = 0
        src[2040:2056, 100:8000] = 0  # across a band seam (tile rows 3 | 4)
        src[8000:8192, 8100:8192] = 0
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=512, path="terrains/streamed", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    roots = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods)), bt.AssetServer().insert("src", src), atlas,
                            defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["bands"] == 4 and st["banded_launches"] == 1 and st["early_tiles"] == 256
            assert st["uploaded_bytes"] == size * size * 2 and st["saved_bytes"] == 341 * 512 * 512 * 2
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        roots.append((root, atlas))
    d0, d1 = (a.attachment_directory(r, 0) for r, a in roots)
    names = sorted(os.listdir(d0))
    assert names == sorted(os.listdir(d1)) and len(names) == 341
    for n in names:
        assert open(os.path.join(d0, n), "rb").read() == open(os.path.join(d1, n), "rb").read(), n
    assert open(os.path.join(roots[0][0], "terrains/streamed/config.tc"), "rb").read() == open(os.path.join(roots[1][0], "terrains/streamed/config.tc"), "rb").read()
    oracle = O.OracleAtlas(lods, 512, False, [(512, 2, 1, O.FORMAT_R16)])
    oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(roots[1][1], oracle) == 341


def test_streamed_run_falls_back_for_queues_it_cannot_band(device, tmp_path):
    """a device-resident raster, a generic-plan job and a cube job are not streamable: the same call still runs and saves"""
    src = K.random_raster(O.FORMAT_R16, 300, 300, seed=93, holes=0.05)
    cfg = bt.TerrainConfig(lod_count=3, atlas_size=64, path="terrains/fallback", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=64, border_size=2, format=bt.AttachmentFormat.R16))
    for generic, defer in ((False, False), (True, True), (False, True)):
        root = str(tmp_path / f"r{int(generic)}{int(defer)}")
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, 3)), bt.AssetServer().insert("src", src), atlas, defer_upload=defer)
        st = pre.run_streamed(atlas, root, generic=generic)
        assert st["streamed"] in (False, True)
        files = os.listdir(atlas.attachment_directory(root, 0))
        assert len(files) == 21
        oracle = K.oracle_planar(src, 3, 64, 2, O.FORMAT_R16)
        for c, i in oracle.tiles():
            data = np.fromfile(os.path.join(atlas.attachment_directory(root, 0), f"{c[0]}_{c[1]}_{c[2]}_{c[3]}.bin"), dtype=np.uint16).reshape(64, 64)
            assert np.array_equal(data, oracle.tile(0, i)), c


def test_config5_cube_albedo_full_size(device):
    """BASELINE config 5's second attachment at full size: 6 cube faces of 8192^2 Rgba8 (albedo), lod_count 5, T = 512 -> 2046
    tiles of 1 MiB through the plan the product picks (fused_direct + tail + cube seams), every tile compared with the
    reference's own WGSL executed on the CPU (oracle/_ref)."""
    W, lods = 8192, 5
    faces = []
    for s in range(6):
        h = K.smooth_raster(W, W, seed=17 + s, device=device)
        rgba = np.empty((W, W, 4), np.uint8)
        rgba[..., 0] = np.maximum(h >> 8, 1)
        rgba[..., 1] = h & 255
        rgba[..., 2] = (h >> 3) & 255
        rgba[..., 3] = 255
        rgba[3000 + 41 * s:3100 + 41 * s, 2000:2400, 0] = 0  # a no-data patch (red = 0) on every face
        faces.append(rgba)
        del h
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=512, border_size=2, format=bt.AttachmentFormat.Rgba8))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    paths = [f"albedo{s}" for s in range(6)]
    for p, f in zip(paths, faces):
        server.insert(p, f)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
    pre.run(atlas)
    st = pre.stats()
    assert st["fused_jobs"] >= 1 and st["tiles"] == 2046
    oracle = K.reference_kernels(O.OracleAtlas(lods, 2048, True, [(512, 2, 1, O.FORMAT_RGBA8)]))
    oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
    assert [((c.side, c.lod, c.x, c.y), i) for c, i in atlas.tiles()] == oracle.tiles()
    for first in range(0, 2046, 64):
        count = min(64, 2046 - first)
        data = atlas.download_tiles(0, first, count)
        for k in range(count):
            assert np.array_equal(data[k], oracle.tile(0, first + k)), (first + k, oracle.tiles()[first + k])


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
@pytest.mark.parametrize("T, b, holes", [(20, 2, 0.0), (44, 4, 0.03), (30, 1, 0.0)])
def test_reference_dispatch_flag_reproduces_the_unwritten_rows(device, fmt, T, b, holes):
    """BT_RUN_REFERENCE_DISPATCH: for a texture size that is not a multiple of 8 the reference dispatches texture_size / 8
    workgroup rows (gpu_tile_atlas.rs:105) and never writes the last texture_size % 8 rows of a tile; stitch then copies
    those unwritten rows into the neighbours' top aprons.  With the flag the product's tiles are the executed WGSL's byte for
    byte (all three shaders, 3 LODs); without it every row is processed (the default, DESIGN.md section 2 finding 1)."""
    lods = 3
    src = K.random_raster(fmt, 150, 170, 77 + T, holes)
    ref = K.reference_kernels(O.OracleAtlas(lods, 128, False, [(T, b, 1, fmt)]))
    ref.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(4)
    atlas, _ = K.product_planar(device, src, lods, T, b, fmt, reference_dispatch=True)
    assert K.assert_atlas_equal(atlas, ref) == 21
    covered = T // 8 * 8
    data = atlas.download_tiles(0, 0, 21)
    assert not data[:, covered:].any()  # the rows nobody wrote still hold the cleared atlas
    full, _ = K.product_planar(device, src, lods, T, b, fmt)
    assert full.download_tiles(0, 0, 21)[:, covered:].any()


@pytest.mark.gpu
def test_ctx_trim_gives_back_the_kept_buffers(device, tmp_path):
    """bt_ctx_trim: a finished queue's device raster (kept for the next queue) and the pinned staging buffers of the save path are
    released, and the next job allocates them again (same files)."""
    src = K.smooth_raster(1024, 1024, seed=5)
    cfg = bt.TerrainConfig(lod_count=3, atlas_size=64, path="terrains/trim", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=256, border_size=2, format=bt.AttachmentFormat.R16))
    digests = []
    for rep in range(2):
        root = str(tmp_path / f"run{rep}")
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer().insert("src", src)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root).preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, 3)), server, atlas)
        pre.run(atlas)
        pre.save(atlas, root)
        pre.close()
        d = atlas.attachment_directory(root, 0)
        digests.append({f: open(os.path.join(d, f), "rb").read() for f in sorted(os.listdir(d))})
        freed = device.trim()
        assert freed >= src.nbytes, freed  # the raster + three pinned buffers
        assert device.trim() == 0
    assert len(digests[0]) == 21 and digests[0] == digests[1]


@pytest.mark.parametrize("W", [2048, 2200, 1016])
def test_dma_variant_fixes_nodata_in_the_chunk_loop_incl_overlays(device, W):
    """Round 5: the LDS-DMA variant of fused_main (T = 512, raster base and pitch 16-byte aligned) handles no-data where it meets it —
    per-pixel validity from the rows still staged, the PREVIOUS atlas texel for pixels without data (split.wgsl:34-42), valid-averages
    for the two parent LODs — instead of redoing flagged chunks behind the loop.  Two datasets over the same tiles: the first leaves
    non-zero previous values, the second has single no-data texels, cell-shaped holes, a band that blanks whole 8-row chunks, holes on tile
    edges (aprons are pulled through the neighbour's formula) and a hole in the first source rows.  Source / tile ratios: ~1 (the static
    9-row path with the occasional skipped row), 1.08 (rows skip often: the table-driven path) and 0.5 (magnification)."""
    T, b, lods = 512, 2, 3
    rng = np.random.default_rng(W)
    base = K.random_raster(O.FORMAT_R16, W, W, seed=W + 1, holes=0.01)
    over = K.random_raster(O.FORMAT_R16, W, W, seed=W + 2)
    over[rng.integers(0, W, 400), rng.integers(0, W, 400)] = 0              # single texels
    for _ in range(40):                                                     # cells like the masked 16k job's
        y, x = rng.integers(0, W - 40), rng.integers(0, W - 60)
        over[y:y + 37, x:x + 53] = 0
    over[W // 3:W // 3 + 30, :] = 0                                         # whole chunks without data
    over[:, W // 2 - 3:W // 2 + 3] = 0                                      # across the x seam of the finest tiles (pulled aprons)
    over[W // 2 - 2:W // 2 + 2, W // 5:W // 2] = 0                          # ... and along a y seam
    over[0:3, 100:300] = 0                                                  # the clamped first rows
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=32, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("base", base).insert("over", over)
    pre = bt.Preprocessor.new()
    pre.preprocess_tile(bt.PreprocessDataset(path="base", lod_range=range(0, lods)), server, atlas).run(atlas)
    oracle = O.OracleAtlas(lods, 32, False, [(T, b, 1, O.FORMAT_R16)])
    oracle.preprocess_tile(0, base, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(atlas, oracle) == 21
    pre.preprocess_tile(bt.PreprocessDataset(path="over", lod_range=range(0, lods)), server, atlas)
    pre.run(atlas, keep_queue=True)
    assert pre.stats()["fused_jobs"] == 1
    oracle.preprocess_tile(0, over, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(atlas, oracle) == 21
    pre.run(atlas)  # the same queue over its own output: every kept texel is now the value it already holds
    assert K.assert_atlas_equal(atlas, oracle) == 21


@pytest.mark.parametrize("W", [1100, 1101, 1104])
def test_unaligned_rasters_are_padded_for_the_16_byte_staging(device, W):
    """Round 5: fused_main moves the source in 16-byte pieces, which needs base and pitch 16-byte aligned.  A host raster the library uploads
    is padded on the way (W = 1100: pitch 2200 -> 2208); a BORROWED device raster with such a pitch is copied into a padded buffer by the
    queue's first run (the caller's memory is read then, not at preprocess_tile); an odd width (1101: pitch 2202) pads the same way; 1104 needs
    nothing.  All against the oracle, with no-data, through the fused plan at T = 512; a kept queue runs again from the copy."""
    T, b, lods = 512, 2, 2
    src = K.random_raster(O.FORMAT_R16, 1000, W, seed=W, holes=0.02)
    oracle = K.oracle_planar(src, lods, T, b, O.FORMAT_R16, atlas_size=16)
    for on_device in (False, True):
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=16, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
        atlas = bt.TileAtlas.new(cfg, device)
        ptr = None
        if on_device:
            ptr = device.upload(np.zeros_like(src))  # tightly packed rows: pitch = 2 W; filled only AFTER the raster has been handed over
            server = bt.AssetServer().insert("s", (ptr, W, 1000))
        else:
            server = bt.AssetServer().insert("s", src)
        pre = bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, lods)), server, atlas)
        if on_device:
            import ctypes
            bt._ffi.check(bt._ffi.lib().bt_memcpy_h2d(device._h, ctypes.c_void_p(ptr), src.ctypes.data_as(ctypes.c_void_p), src.nbytes))
            device.synchronize()
        pre.run(atlas, keep_queue=True)
        assert pre.stats()["fused_jobs"] == 1
        assert K.assert_atlas_equal(atlas, oracle) == 5, on_device
        pre.run(atlas)
        assert K.assert_atlas_equal(atlas, oracle) == 5, on_device
        if ptr is not None:
            device.free(ptr)


def test_streamed_run_pads_a_deferred_raster_of_unaligned_pitch(device, tmp_path):
    """A deferred host raster whose rows are not 16-byte aligned (4100 texels: pitch 8200) gets a padded device copy too: its bands travel
    as pitched copies, fused_main stages it by LDS-DMA, and the streamed run's files equal the serial run's and the oracle's tiles."""
    W, lods = 4100, 4
    src = K.random_raster(O.FORMAT_R16, W, W, seed=12, holes=0.01)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=128, path="terrains/streamed_u", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    roots = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods)), bt.AssetServer().insert("src", src), atlas,
                            defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["bands"] == 4  # (8 tile rows: bands of 2)
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        roots.append((root, atlas))
    d0, d1 = (a.attachment_directory(r, 0) for r, a in roots)
    names = sorted(os.listdir(d0))
    assert names == sorted(os.listdir(d1)) and len(names) == 85
    for n in names:
        assert open(os.path.join(d0, n), "rb").read() == open(os.path.join(d1, n), "rb").read(), n
    assert K.assert_atlas_equal(roots[1][1], K.oracle_planar(src, lods, 512, 2, O.FORMAT_R16, atlas_size=128)) == 85


@pytest.mark.parametrize("shape", [(1, 1), (3, 5), (9, 1), (1, 17), (8, 8)])
@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_tiny_rasters_through_the_512_wide_plans(device, fmt, shape):
    """Extreme magnification at the production tile size: a raster of a few texels (padded to 16-byte rows by the library) blown up to
    5 tiles of 512^2 — every source index clamps, whole chunks read ONE source row — through the plan the product picks and the generic one;
    with a no-data texel where there is room for one."""
    src = K.random_raster(fmt, shape[0], shape[1], seed=shape[0] * 31 + shape[1])
    if src.shape[0] * src.shape[1] > 4:
        (src[..., 0] if fmt == O.FORMAT_RGBA8 else src)[src.shape[0] // 2, src.shape[1] // 2] = 0
    oracle = K.oracle_planar(src, 2, 512, 2, fmt, atlas_size=8)
    for generic in (False, True):
        atlas, _ = K.product_planar(device, src, 2, 512, 2, fmt, atlas_size=8, generic=generic)
        assert K.assert_atlas_equal(atlas, oracle) == 5, generic


def test_32k_job_seven_lods_beyond_the_bench_size(device):
    """Maximum-size check one step past BASELINE's largest input: a 32768^2 R16 source (2 GiB: row and layer offsets beyond 2^31 bytes),
    lod_count 7 -> 5461 tiles (a 2.7 GB atlas attachment: texel offsets up to 1.4 x 10^9), a no-data patch across a tile seam.  The index
    contract for all 5461 tiles, and every 11th tile + the whole top of the pyramid byte for byte against the oracle."""
    size, lods = 32768, 7
    ptr = device.synth_fbm_r16(size, size, 77)
    src = device.download(ptr, (size, size), np.uint16)
    src[16000:16700, 20000:20900] = 0
    device.free(ptr)
    ptr = device.upload(src)
    n_tiles = sum(4 ** l for l in range(lods))
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=5500, path="terrains/big", model=bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
        bt.PreprocessDataset(attachment_index=0, path="big", lod_range=range(0, lods)), bt.AssetServer().insert("big", (ptr, size, size)), atlas)
    pre.run(atlas)
    device.free(ptr)
    st = pre.stats()
    assert st["fused_jobs"] == 1 and st["tiles"] == n_tiles
    oracle = O.OracleAtlas(lods, 5500, False, [(512, 2, 1, O.FORMAT_R16)])
    oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(O.usable_cores())
    assert [((c.side, c.lod, c.x, c.y), i) for c, i in atlas.tiles()] == oracle.tiles()
    picked = sorted(set(range(0, n_tiles, 11)) | set(range(n_tiles - 341, n_tiles)))  # a sample of the two finest LODs + LODs 0 .. 4 completely
    for i in picked:
        assert np.array_equal(atlas.download_tiles(0, i, 1)[0], oracle.tile(0, i)), (i, oracle.tiles()[i])


def _files_equal(dir_a, dir_b, count):
    names = sorted(os.listdir(dir_a))
    assert names == sorted(os.listdir(dir_b)) and len(names) == count, (len(names), count)
    for n in names:
        assert open(os.path.join(dir_a, n), "rb").read() == open(os.path.join(dir_b, n), "rb").read(), n


@pytest.mark.parametrize("holes", [False, True])
def test_streamed_run_two_attachments_like_preprocess_planar(device, tmp_path, holes):
    """examples/preprocess_planar.rs:16-60 end to end: height (R16) + albedo (Rgba8) of 4096^2 in ONE queue, both rasters handed over
    deferred.  The streamed pipeline bands fused_main AND fused_direct (4 bands of 2 tile rows each), interleaves the two attachments'
    uploads, kernels and saves; its files are byte-identical to run() + save() and its atlas to the oracle's (both attachments)."""
    W, lods = 4096, 4
    height = K.smooth_raster(W, W, seed=1234, device=device)
    rng = np.random.default_rng(1235)
    albedo = rng.integers(1, 256, size=(W, W, 4), dtype=np.uint8)
    albedo[..., 3] = 255
    if holes:
        height[1000:1100, 500:2500] = 0  # across the band seam between tile rows 1 | 2 (mosaic row 1016)
        height[4000:4096, 0:300] = 0
        albedo[2020:2050, 100:4000, 0] = 0  # across tile rows 3 | 4
        albedo[0:40, 4000:4096, 0] = 0
    cfg = bt.TerrainConfig(lod_count=lods, path="terrains/planar", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=512, border_size=2, format=bt.AttachmentFormat.Rgba8))
    server = bt.AssetServer().insert("h", height).insert("a", albedo)
    roots = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root).clear_attachment(1, atlas, root)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=1, path="a", lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["banded_launches"] == 2 and st["bands"] == 8 and st["early_tiles"] == 128
            assert st["uploaded_bytes"] == W * W * 6 and st["saved_bytes"] == 85 * 512 * 512 * 6
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        roots.append((root, atlas))
    for ai in (0, 1):
        _files_equal(roots[0][1].attachment_directory(roots[0][0], ai), roots[1][1].attachment_directory(roots[1][0], ai), 85)
    assert open(os.path.join(roots[0][0], "terrains/planar/config.tc"), "rb").read() == open(os.path.join(roots[1][0], "terrains/planar/config.tc"), "rb").read()
    oracle = O.OracleAtlas(lods, 1024, False, [(512, 2, 1, O.FORMAT_R16), (512, 2, 1, O.FORMAT_RGBA8)])
    oracle.clear_attachment(0).clear_attachment(1)
    oracle.preprocess_tile(0, height, (0, lods)).preprocess_tile(1, albedo, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(roots[1][1], oracle, 0) == 85
    assert K.assert_atlas_equal(roots[1][1], oracle, 1) == 85


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_streamed_run_cube_job_like_preprocess_spherical(device, tmp_path, fmt):
    """examples/preprocess_spherical.rs:20-48 end to end: six deferred face rasters (2048^2, T = 512, lod_count 3 -> 4 x 4 finest tiles
    per face, 126 tiles).  A face's bands upload while the previous face's kernels and downloads run; the finest tiles on a face edge
    wait for the seam stitch behind the last face (12 of 16 per face), the interior ones leave with their band.  Files byte-identical
    to the serial path's, atlas identical to the oracle's."""
    W, lods, T = 2048, 3, 512
    faces = []
    for s in range(6):
        h = K.smooth_raster(W, W, seed=300 + s, device=device)
        if fmt == O.FORMAT_R16:
            h[500 + 37 * s:560 + 37 * s, 0:700] = 0  # no data up to a face edge (the seam copies what the neighbour's centre holds)
            faces.append(h)
        else:
            rgba = np.empty((W, W, 4), np.uint8)
            rgba[..., 0] = np.maximum(h >> 8, 1)
            rgba[..., 1] = h & 255
            rgba[..., 2] = (h >> 3) & 255
            rgba[..., 3] = 255
            rgba[1500:1530, 1200 + 11 * s:2048, 0] = 0
            faces.append(rgba)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=256, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=2, format=K.FMT[fmt]))
    server = bt.AssetServer()
    paths = [f"face{s}" for s in range(6)]
    for pth, f in zip(paths, faces):
        server.insert(pth, f)
    roots = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root).preprocess_spherical(
            bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["banded_launches"] == 1 and st["bands"] == 24 and st["early_tiles"] == 6 * 4
            px = 2 if fmt == O.FORMAT_R16 else 4
            assert st["uploaded_bytes"] == 6 * W * W * px and st["saved_bytes"] == 126 * T * T * px
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        roots.append((root, atlas))
    _files_equal(roots[0][1].attachment_directory(roots[0][0], 0), roots[1][1].attachment_directory(roots[1][0], 0), 126)
    assert open(os.path.join(roots[0][0], "terrains/spherical/config.tc"), "rb").read() == open(os.path.join(roots[1][0], "terrains/spherical/config.tc"), "rb").read()
    oracle = O.OracleAtlas(lods, 256, True, [(T, 2, 1, fmt)])
    oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(roots[1][1], oracle) == 126


def test_streamed_run_bands_what_it_can_and_runs_the_rest_whole(device, tmp_path):
    """One queue, two jobs: the first raster is handed over deferred (its fused_main launch is banded, its finest tiles leave early), the
    second was uploaded by preprocess_tile (nothing to stream: its launch runs whole, in plan order, its tiles leave behind it).
    Files == serial path, atlas == oracle."""
    W, lods = 2048, 3
    a_src = K.smooth_raster(W, W, seed=5, device=device)
    b_src = K.smooth_raster(W, W, seed=6, device=device)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=64, path="terrains/mixed", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="one", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    cfg.add_attachment(bt.AttachmentConfig(name="two", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    server = bt.AssetServer().insert("a", a_src).insert("b", b_src)
    roots = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root).clear_attachment(1, atlas, root)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="a", lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=1, path="b", lod_range=range(0, lods)), server, atlas, defer_upload=False)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["banded_launches"] == 1 and st["bands"] == 4 and st["early_tiles"] == 16
            assert st["uploaded_bytes"] == W * W * 2 and st["saved_bytes"] == 2 * 21 * 512 * 512 * 2
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        roots.append((root, atlas))
    for ai in (0, 1):
        _files_equal(roots[0][1].attachment_directory(roots[0][0], ai), roots[1][1].attachment_directory(roots[1][0], ai), 21)
    oracle = O.OracleAtlas(lods, 64, False, [(512, 2, 1, O.FORMAT_R16), (512, 2, 1, O.FORMAT_R16)])
    oracle.clear_attachment(0).clear_attachment(1).preprocess_tile(0, a_src, (0, lods)).preprocess_tile(1, b_src, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(roots[1][1], oracle, 0) == 21 and K.assert_atlas_equal(roots[1][1], oracle, 1) == 21


def test_streamed_run_of_an_overlay_keeps_its_tiles_until_the_last_job(device, tmp_path):
    """Two datasets onto the SAME tiles of one attachment (the second overlays the first where it has data): a tile must not leave
    with the first job's band — the second job writes it again.  Early saves are off for such an attachment; files == serial path."""
    W, lods = 2048, 3
    base = K.smooth_raster(W, W, seed=11, device=device)
    over = K.smooth_raster(W, W, seed=12, device=device)
    over[:, 1000:] = 0
    over[300:900, 200:700] = 0
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=64, path="terrains/overlay", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    server = bt.AssetServer().insert("base", base).insert("over", over)
    roots = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
        for name in ("base", "over"):
            pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path=name, lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["banded_launches"] == 2 and st["early_tiles"] == 0
            assert pre.stats()["prev_zero_launches"] == 1  # the first job only: the second finds its tiles written
        else:
            pre.run(atlas)
            assert pre.stats()["prev_zero_launches"] == 1
            pre.save(atlas, root)
        roots.append((root, atlas))
    _files_equal(roots[0][1].attachment_directory(roots[0][0], 0), roots[1][1].attachment_directory(roots[1][0], 0), 21)
    oracle = O.OracleAtlas(lods, 64, False, [(512, 2, 1, O.FORMAT_R16)])
    oracle.clear_attachment(0).preprocess_tile(0, base, (0, lods)).preprocess_tile(0, over, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(roots[1][1], oracle) == 21


@pytest.mark.parametrize("fmt,T,W,lods", [(O.FORMAT_R16, 512, 2048, 3), (O.FORMAT_R16, 64, 520, 4), (O.FORMAT_RGBA8, 512, 2048, 3), (O.FORMAT_RGBA8, 36, 300, 4)])
def test_fresh_atlas_takes_the_previous_value_as_zero_without_fetching(device, fmt, T, W, lods):
    """A job onto layers nothing has written since bt_atlas_create runs with FusedArgs::prev_zero (no-data pixels become 0 without the
    fetch of split.wgsl:34-42's previous value): same tiles as the oracle; the re-run of the kept queue, an overlay onto written
    tiles, a tile uploaded by the host and a handed-out storage pointer all take the fetching path — same tiles again."""
    src = K.random_raster(fmt, W, W, seed=T + W, holes=0.04)
    if fmt == O.FORMAT_R16:
        src[W // 3:W // 3 + 40, :] = 0  # whole rows without data, across tile seams
    else:
        src[W // 3:W // 3 + 40, :, 0] = 0
    tiles = sum(4 ** l for l in range(lods))
    oracle = K.oracle_planar(src, lods, T, 2, fmt, atlas_size=128)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=128, path="terrains/test", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=2, format=K.FMT[fmt]))
    ds = bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods))
    server = bt.AssetServer().insert("src", src)
    # fresh: flagged; the kept queue's second run is not (its tiles are written now) and rewrites the same bytes
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(ds, server, atlas)
    pre.run(atlas, keep_queue=True)
    assert pre.stats()["prev_zero_launches"] == 1 and pre.stats()["fused_jobs"] == 1
    assert K.assert_atlas_equal(atlas, oracle) == tiles
    pre.run(atlas)
    assert pre.stats()["prev_zero_launches"] == 0
    assert K.assert_atlas_equal(atlas, oracle) == tiles
    # a host upload into one finest layer of a fresh atlas: not flagged, and the no-data pixels of that tile keep the uploaded texels
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(ds, server, atlas)
    finest = [i for c, i in oracle.tiles() if c[1] == lods - 1]
    marker = np.full((T, T) if fmt == O.FORMAT_R16 else (T, T, 4), 7, np.uint16 if fmt == O.FORMAT_R16 else np.uint8)
    atlas.upload_tile(0, finest[3], marker)
    pre.run(atlas)
    assert pre.stats()["prev_zero_launches"] == 0
    ours = atlas.download_tiles(0, finest[3], 1)[0]
    theirs = oracle.tile(0, finest[3])
    nodata = (theirs == 0) if fmt == O.FORMAT_R16 else (theirs[..., 0] == 0)  # (a blend of texels >= 1 never rounds to 0)
    centre = np.zeros((T, T), bool)
    centre[2:T - 2, 2:T - 2] = True
    assert nodata[centre].any()
    assert (ours[centre & nodata] == 7).all() and np.array_equal(ours[centre & ~nodata], theirs[centre & ~nodata])
    # the storage pointer handed out (the caller may write through it): not flagged either
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(ds, server, atlas)
    atlas.attachment_storage(0)
    pre.run(atlas)
    assert pre.stats()["prev_zero_launches"] == 0
    assert K.assert_atlas_equal(atlas, oracle) == tiles


def test_config5_streamed_at_full_size_writes_the_serial_path_s_files(device, tmp_path):
    """BASELINE config 5's height job at full size (6 faces of 8192^2 R16, lod_count 5, 2046 tiles, 0.8 GB in / 1.07 GB out) through the streamed
    pipeline: 24 bands of 4 tile rows, 1176 interior finest tiles leave with their bands, the 360 face-edge ones and the 510 parents behind
    the seam stitch.  Every file equals the serial path's (whose atlas test_config5_* compares with the executed WGSL)."""
    import hashlib

    W, lods = 8192, 5
    faces = [K.smooth_raster(W, W, seed=7 + s, device=device) for s in range(6)]
    for s in range(6):
        faces[s][4000 + 13 * s:4100 + 13 * s, 0:900] = 0  # no data up to a face edge, across a band seam (tile rows 7 | 8: mosaic row 4064)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
    server = bt.AssetServer()
    paths = [f"face{s}" for s in range(6)]
    for pth, f in zip(paths, faces):
        server.insert(pth, f)
    digests = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root).preprocess_spherical(
            bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["bands"] == 24 and st["early_tiles"] == 6 * 14 * 14
            assert st["uploaded_bytes"] == 6 * W * W * 2 and st["saved_bytes"] == 2046 * 512 * 512 * 2
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        d = atlas.attachment_directory(root, 0)
        names = sorted(os.listdir(d))
        assert len(names) == 2046
        digests.append({n: hashlib.sha256(open(os.path.join(d, n), "rb").read()).hexdigest() for n in names})
        digests[-1]["config.tc"] = hashlib.sha256(open(os.path.join(root, "terrains/spherical/config.tc"), "rb").read()).hexdigest()
        atlas.close()
        import shutil

        shutil.rmtree(root, ignore_errors=True)
    assert digests[0] == digests[1]


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
@pytest.mark.parametrize("T,b,lods,W", [(20, 2, 6, 330), (36, 4, 6, 500), (16, 2, 7, 400), (64, 2, 5, 900)])
def test_cube_seams_of_the_top_lods_are_pulled_inside_the_tail_launch(device, T, b, lods, W, fmt):
    """Round 6: the cross-face apron regions of the LODs the tail launch itself produces are evaluated from the tail's INPUT on the neighbour
    face (one, two or three LODs up) instead of copied by a stitch launch behind it: lod_count 6 = three LODs below fused_main's (a 64-pixel
    input block per LOD-0 apron pixel, the recursive form), lod_count 5 = two (one tile lookup per pixel), lod_count 7 = two tail launches (the
    plan then keeps the stitch launch).  Every tile == the oracle's, with no-data up to the face edges; the launch count says which plan ran."""
    faces = [K.random_raster(fmt, W, W, seed=900 + 7 * s + T, holes=0.03) for s in range(6)]
    for s in range(6):  # no data along a whole face edge: the pulled texels are valid-averages of partly empty blocks, or 0
        if fmt == O.FORMAT_R16:
            faces[s][: W // 9, :] = 0
            faces[s][:, W - W // 11:] = 0
        else:
            faces[s][: W // 9, :, 0] = 0
            faces[s][:, W - W // 11:, 0] = 0
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=6 * sum(4 ** l for l in range(lods)) + 8, path="terrains/pull")
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b, format=K.FMT[fmt]))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    paths = [f"f{s}" for s in range(6)]
    for pth, f in zip(paths, faces):
        server.insert(pth, f)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
    pre.run(atlas, keep_queue=True)
    st = pre.stats()
    assert st["fused_jobs"] == 1 and st["kernel_launches"] == (2 if lods <= 6 else 4), st  # main / direct + tail (+ a second tail + the stitch launch for 7 LODs)
    oracle = O.OracleAtlas(lods, cfg.atlas_size, True, [(T, b, 1, fmt)])
    oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
    n = 6 * sum(4 ** l for l in range(lods))
    assert K.assert_atlas_equal(atlas, oracle) == n
    pre.run(atlas)  # onto the written atlas (previous values fetched): the same tiles
    assert K.assert_atlas_equal(atlas, oracle) == n


def test_streamed_run_cube_with_two_attachments_like_the_spherical_example(device, tmp_path):
    """examples/preprocess_spherical.rs:20-48 in full: height (R16) AND albedo (Rgba8) of a six-face cube in one queue, all twelve rasters deferred.
    fused_main's and fused_direct's bands (24 + 24) interleave with the saves of both attachments; files == the serial path's, atlas == the oracle's."""
    W, lods, T = 1024, 3, 256
    heights = [K.smooth_raster(W, W, seed=500 + s, device=device) for s in range(6)]
    albedos = []
    for s, h in enumerate(heights):
        rgba = np.empty((W, W, 4), np.uint8)
        rgba[..., 0] = np.maximum(h >> 8, 1)
        rgba[..., 1] = h & 255
        rgba[..., 2] = (h >> 5) & 255
        rgba[..., 3] = 255
        rgba[700:760, 0:300 + 20 * s, 0] = 0
        albedos.append(rgba)
        h[100 + 9 * s:140 + 9 * s, 600:W] = 0
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=256, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=2, format=bt.AttachmentFormat.R16))
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=T, border_size=2, format=bt.AttachmentFormat.Rgba8))
    server = bt.AssetServer()
    hp, ap = [f"h{s}" for s in range(6)], [f"a{s}" for s in range(6)]
    for s in range(6):
        server.insert(hp[s], heights[s]).insert(ap[s], albedos[s])
    roots = []
    for streamed in (False, True):
        root = str(tmp_path / ("streamed" if streamed else "serial"))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root).clear_attachment(1, atlas, root)
        pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=hp, lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        pre.preprocess_spherical(bt.SphericalDataset(attachment_index=1, paths=ap, lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            assert st["streamed"] and st["banded_launches"] == 2 and st["bands"] == 48 and st["early_tiles"] == 2 * 6 * 4
            assert st["uploaded_bytes"] == 6 * W * W * 6 and st["saved_bytes"] == 126 * T * T * 6
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        roots.append((root, atlas))
    for ai in (0, 1):
        _files_equal(roots[0][1].attachment_directory(roots[0][0], ai), roots[1][1].attachment_directory(roots[1][0], ai), 126)
    oracle = O.OracleAtlas(lods, 256, True, [(T, 2, 1, O.FORMAT_R16), (T, 2, 1, O.FORMAT_RGBA8)])
    oracle.clear_attachment(0).clear_attachment(1)
    oracle.preprocess_spherical(0, heights, (0, lods)).preprocess_spherical(1, albedos, (0, lods)).run(O.usable_cores())
    assert K.assert_atlas_equal(roots[1][1], oracle, 0) == 126 and K.assert_atlas_equal(roots[1][1], oracle, 1) == 126


@pytest.mark.parametrize("kind", ["planar_r16", "planar_rgba8", "cube_r16", "cube_rgba8"])
def test_tiles_in_layers_beyond_2_to_32_texels(device, kind):
    """An attachment of more than 2^32 texels (16384 tiles of 512^2; GEBCO-scale terrains have 21845+): the job's tiles sit in layers 16400 and up,
    where layer x tile texels no longer fits 32 bits.  Rounds 2 - 6 sent such atlases to the batched kernels (4 x slower); now they take the fused
    plans — every tile byte for byte against the oracle at (index - 16400), and the layers a wrapped offset would hit (index mod 16384) still zero."""
    shift, T, b = 16400, 512, 2
    fmt = O.FORMAT_RGBA8 if kind.endswith("rgba8") else O.FORMAT_R16
    cube = kind.startswith("cube")
    lods = 2 if cube else 3
    n = 1100 if cube else 2200
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=shift + 40, path="terrains/high", **({} if cube else dict(model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
    atlas = bt.TileAtlas.new(cfg, device)
    for i in range(shift):  # occupy the first 16400 layers (coordinates no job touches)
        atlas.get_or_allocate_tile(bt.TileCoordinate(0, 30, i, 0))
    server = bt.AssetServer()
    oracle = O.OracleAtlas(lods, 64, cube, [(T, b, 1, fmt)])
    pre = bt.Preprocessor.new().clear_attachment(0, atlas)
    if cube:
        faces = [K.random_raster(fmt, n, n, 900 + s, holes=0.02) for s in range(6)]
        paths = [f"f{s}" for s in range(6)]
        for path, f in zip(paths, faces):
            server.insert(path, f)
        pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(16)
    else:
        src = K.random_raster(fmt, n, n, 901, holes=0.02)
        server.insert("src", src)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods)), server, atlas)
        oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(16)
    pre.run(atlas, keep_queue=True)
    st = pre.stats()
    assert st["fused_jobs"] == 1 and st["generic_jobs"] == 0, st
    expected = oracle.tiles()
    for rerun in (False, True):  # fresh (previous values taken as zero) and onto the written tiles (previous values fetched)
        if rerun:
            pre.run(atlas)
        ours = [((c.side, c.lod, c.x, c.y), i - shift) for c, i in atlas.tiles() if c.lod < 30]
        assert ours == expected
        data = atlas.download_tiles(0, shift, len(expected))
        bad = [coord for coord, idx in expected if not np.array_equal(data[idx], oracle.tile(0, idx))]
        assert not bad, (rerun, len(bad), bad[:4])
        low = atlas.download_tiles(0, 0, 96)
        assert not low.any(), "a wrapped 32-bit offset wrote into the first layers"
        assert not atlas.download_tiles(0, shift + len(expected), 2).any()


@pytest.mark.parametrize("ratio,cube,fmt", [(1.23, False, O.FORMAT_R16), (1.45, False, O.FORMAT_R16), (0.8, False, O.FORMAT_R16), (1.3, True, O.FORMAT_R16),
                                            (1.41, False, O.FORMAT_R16), (1.8, False, O.FORMAT_R16), (1.5, True, O.FORMAT_R16),
                                            (0.71, False, O.FORMAT_RGBA8), (0.93, False, O.FORMAT_RGBA8), (1.04, False, O.FORMAT_RGBA8), (1.2, False, O.FORMAT_RGBA8),
                                            (1.41, False, O.FORMAT_RGBA8), (1.52, False, O.FORMAT_RGBA8), (1.3, True, O.FORMAT_RGBA8), (0.8, True, O.FORMAT_RGBA8)])
def test_source_to_tile_ratios_staged_by_dma_alone(device, tmp_path, ratio, cube, fmt):
    """T = 512 with a source that is not the size of the tile mosaic (real datasets rarely are: GEBCO's 86400 columns over 128 x 508): from a ratio of
    ~1.2 the staged window has more 16-byte pieces than the register staging batches, and rounds 2 - 6 then ran the unstaged kernel (1.9 TB/s where the
    16k job runs at 4.3).  Such windows are now staged by LDS-DMA with a run-time pitch (4.2 TB/s at ratio 1.23) — with the apron rows from global memory
    (ratios ~1.25 - 1.35) or ONE staging buffer (from ~1.36) where that keeps four workgroups on a CU: every tile against the oracle — fresh, onto the written
    atlas (no-data texels fetch their previous value) and through the streamed pipeline.
    Rgba8 (fused_direct): its requests-ahead path took only blocks whose rows step through the source one row at a time; now also rows that repeat the pair
    above (ratios below 1) and, in a variant of its own (ratios above 1.02), rows that pass over a source row — up to two per block (1.52 is past that), the passed-to rows by plain loads."""
    T, b, lods = 512, 2, 2 if cube else 3
    n = int(((T - 2 * b) << (lods - 1)) * ratio)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=64, path="terrains/ratio", **({} if cube else dict(model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
    oracle = O.OracleAtlas(lods, 64, cube, [(T, b, 1, fmt)])
    server = bt.AssetServer()
    if cube:
        faces = [K.random_raster(fmt, n, n, 70 + s, holes=0.03) for s in range(6)]
        paths = [f"f{s}" for s in range(6)]
        for path, f in zip(paths, faces):
            server.insert(path, f)
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(16)
    else:
        src = K.random_raster(fmt, n, n, 71, holes=0.03)
        server.insert("src", src)
        oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(16)

    def queue(atlas, root=None, defer=False):
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
        if cube:
            return pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas, defer_upload=defer)
        return pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods)), server, atlas, defer_upload=defer)

    atlas = bt.TileAtlas.new(cfg, device)
    pre = queue(atlas)
    pre.run(atlas, keep_queue=True)
    assert pre.stats()["fused_jobs"] == 1
    n_tiles = K.assert_atlas_equal(atlas, oracle)
    pre.run(atlas)
    assert K.assert_atlas_equal(atlas, oracle) == n_tiles
    atlas2 = bt.TileAtlas.new(cfg, device)
    queue(atlas2, str(tmp_path), defer=True).run_streamed(atlas2, str(tmp_path))
    assert K.assert_atlas_equal(atlas2, oracle) == n_tiles
    assert len(os.listdir(atlas2.attachment_directory(str(tmp_path), 0))) == n_tiles
