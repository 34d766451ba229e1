"""Seeded random sweep over the job parameters the launch planner branches on (tile shape, border, LOD count, format,
raster shape, holes, dataset rectangle, lod_range offset, planar / cube): whatever plan the product picks — fused,
direct, hybrid or generic — every tile must equal the oracle's, byte for byte, at the same atlas index."""
import os

import numpy as np
import pytest

import _cases as K
import _oracle as O
import bevy_terrain_amd as bt

pytestmark = pytest.mark.gpu
FUZZ = int(os.environ.get("BT_FUZZ_OFFSET", "0"))  # other seeds of the same sweep: BT_FUZZ_OFFSET=1000 pytest -m gpu -k ...


@pytest.fixture(scope="module")
def device():
    yield bt.Device(0)


def draw_case(seed):
    rng = np.random.default_rng(10_000 + seed)
    T = int(rng.choice([12, 16, 20, 24, 36, 40, 64, 100, 128, 132, 256, 260, 512]))
    b = int(rng.choice([x for x in (1, 2, 3, 4, 6, 8) if T - 2 * x >= 2 * x and T - 2 * x >= 4]))
    max_lods = 5 if T <= 64 else 4 if T <= 132 else 3
    lods = int(rng.integers(1, max_lods + 1))
    fmt = O.FORMAT_R16 if rng.random() < 0.5 else O.FORMAT_RGBA8
    ratio = float(rng.choice([0.3, 0.9, 1.0, 1.3, 2.2, 4.5]))  # source texels per finest tile texel (up / down sampling)
    extent = (T - 2 * b) << (lods - 1)
    W = int(np.clip(extent * ratio * rng.uniform(0.8, 1.2), 8, 3000))
    H = int(np.clip(extent * ratio * rng.uniform(0.8, 1.2), 8, 3000))
    holes = float(rng.choice([0.0, 0.0, 0.02, 0.3]))
    ds = {}
    if rng.random() < 0.3:
        x0, y0 = rng.uniform(0.0, 0.5, 2)
        ds = dict(top_left=(float(x0), float(y0)), bottom_right=(float(x0 + rng.uniform(0.2, 0.5)), float(y0 + rng.uniform(0.2, 0.5))))
    cube = bool(rng.random() < 0.15 and T <= 132)
    lod_begin = int(rng.integers(0, lods)) if rng.random() < 0.25 else 0
    overlay = bool(rng.random() < 0.2)  # a second dataset over part of the first (keep-previous where it has no data)
    return dict(T=T, b=b, lods=lods, fmt=fmt, W=W, H=H, holes=holes, ds=ds, cube=cube, lod_begin=lod_begin, overlay=overlay)


@pytest.mark.parametrize("seed", range(FUZZ, FUZZ + 160))
def test_random_job_matches_oracle(device, seed, tmp_path):
    p = draw_case(seed)
    T, b, lods, fmt = p["T"], p["b"], p["lods"], p["fmt"]
    if p["cube"]:
        n = min(p["W"], 700)
        faces = [K.random_raster(fmt, n, n, seed * 7 + s, holes=p["holes"]) for s in range(6)]
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2100, path="terrains/sweep")
        cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer()
        paths = [f"f{s}" for s in range(6)]
        for path, f in zip(paths, faces):
            server.insert(path, f)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
            bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
        pre.run(atlas, keep_queue=True)
        oracle = O.OracleAtlas(lods, 2100, True, [(T, b, 1, fmt)])
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(16)
        n_tiles = 6 * sum(4 ** l for l in range(lods))
        assert K.assert_atlas_equal(atlas, oracle) == n_tiles, p
        # round 6: the first run found the atlas unwritten (previous values of no-data pixels taken as 0 without a fetch); the re-run of the kept
        # queue finds it written and FETCHES them — the same tiles; and the streamed pipeline (six deferred faces, band by band) writes the same files
        pre.run(atlas)
        assert K.assert_atlas_equal(atlas, oracle) == n_tiles, (p, "re-run")
        atlas2 = bt.TileAtlas.new(cfg, device)
        pre2 = bt.Preprocessor.new().clear_attachment(0, atlas2, str(tmp_path)).preprocess_spherical(
            bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas2, defer_upload=True)
        pre2.run_streamed(atlas2, str(tmp_path))
        assert K.assert_atlas_equal(atlas2, oracle) == n_tiles, (p, "streamed")
        assert len(os.listdir(atlas2.attachment_directory(str(tmp_path), 0))) == n_tiles
        return
    src = K.random_raster(fmt, p["H"], p["W"], seed, holes=p["holes"])
    over = K.random_raster(fmt, max(p["H"] // 2, 8), max(p["W"] // 3, 8), seed + 5000, holes=0.25)
    over_ds = dict(top_left=(0.25, 0.125), bottom_right=(0.75, 0.5))
    lod_range = (p["lod_begin"], lods)
    oracle = O.OracleAtlas(lods, 400, False, [(T, b, 1, fmt)])
    oracle.clear_attachment(0).preprocess_tile(0, src, lod_range, **p["ds"])
    if p["overlay"]:
        oracle.preprocess_tile(0, over, lod_range, **over_ds)
    oracle.run(16)
    for generic in (False, True):  # whatever plan the product picks, and the reference-shaped batched kernels
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=400, path="terrains/sweep", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer().insert("src", src).insert("over", over)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
            bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(*lod_range), **p["ds"]), server, atlas)
        if p["overlay"]:
            pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="over", lod_range=range(*lod_range), **over_ds), server, atlas)
        pre.run(atlas, generic=generic, keep_queue=not p["overlay"])
        n_tiles = K.assert_atlas_equal(atlas, oracle)
        assert n_tiles > 0, (p, generic)
        if not p["overlay"]:  # the kept queue once more, onto written tiles: no-data pixels now FETCH their previous value (round 6: the first run did not)
            pre.run(atlas, generic=generic)
            assert K.assert_atlas_equal(atlas, oracle) == n_tiles, (p, generic, "re-run")
    # ... and the same queue through the streamed pipeline (deferred rasters, band by band where the plan is a fused one): same atlas, same file count
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas, str(tmp_path)).preprocess_tile(
        bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(*lod_range), **p["ds"]), server, atlas, defer_upload=True)
    if p["overlay"]:
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="over", lod_range=range(*lod_range), **over_ds), server, atlas, defer_upload=True)
    pre.run_streamed(atlas, str(tmp_path))
    assert K.assert_atlas_equal(atlas, oracle) == n_tiles, (p, "streamed")
    assert len(os.listdir(atlas.attachment_directory(str(tmp_path), 0))) == n_tiles
