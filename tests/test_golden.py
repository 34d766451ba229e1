"""Committed golden vectors (tests/golden/*.npz).  Since round 3 they are OUTPUTS OF THE REFERENCE'S OWN WGSL, executed
on the CPU by oracle/wgsl_ref (tests/golden/make_golden.py; formats.npz alone is oracle-made: Rust-side byte formats).
The hand-written oracle must reproduce them (CPU, also on the GPU box where /root/reference does not exist), and the HIP
product must reproduce them through the C ABI without any oracle in the loop (GPU)."""
import glob
from ctypes import sizeof as C_sizeof
import hashlib
import os

import numpy as np
import pytest

import _oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, "*.npz")) if not p.endswith(("formats.npz", "refine.npz")))


def check_tiles(g, tiles_of):
    """tiles_of(atlas_index) -> ndarray; compare with the fixture (texels when stored, digests always)."""
    for row, coord in enumerate(g["coords"]):
        t = np.ascontiguousarray(tiles_of(int(coord[4])))
        assert hashlib.sha256(t.tobytes()).digest() == g["tile_sha256"][row].tobytes(), tuple(coord)
        if "tiles" in g:
            assert np.array_equal(t, g["tiles"][row]), tuple(coord)


def test_fixture_set_is_complete():
    assert CASES == ["cube_r16_t16", "cube_rgba8_t16", "planar_r16_t128_one_hole", "planar_r16_t16", "planar_r16_t24_overlay",
                     "planar_r16_t32_subrect", "planar_r16_t64", "planar_rgba8_t16", "planar_rgba8_t32_b3"]
    for name in CASES + ["refine"]:
        g = np.load(os.path.join(GOLDEN, name + ".npz"))
        assert str(g["generator"]) == "oracle/_ref: executed WGSL"
        assert any("preprocess/split.wgsl" in s for s in g["wgsl_sha256"].tolist())


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_golden(name):
    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    fmt, T, b, lods = (int(v) for v in g["params"])
    cube = name.startswith("cube")
    a = O.OracleAtlas(lods, 128, cube, [(T, b, 1, fmt)])
    a.clear_attachment(0)
    if cube:
        a.preprocess_spherical(0, list(g["source"]), (0, lods))
    else:
        for src, r in zip(g["source"], g["rects"]):
            a.preprocess_tile(0, src, (0, lods), top_left=(float(r[0]), float(r[1])), bottom_right=(float(r[2]), float(r[3])))
    a.run(4)
    assert [list(c) + [i] for c, i in a.tiles()] == g["coords"].tolist()
    check_tiles(g, lambda i: a.tile(0, i))


REFINE_PATHS = ("planar", "sphere", "ellipsoid")


def refine_frames(g, name):
    views, off = g[name + "_views"], g[name + "_offsets"]
    for k in range(len(views)):
        yield k, O.View.from_buffer_copy(views[k].tobytes()), g[name + "_tiles"][off[k]:off[k + 1]], g[name + "_indirect"][k].tolist(), g[name + "_passes"][k].tolist()


@pytest.mark.parametrize("name", REFINE_PATHS)
def test_oracle_reproduces_golden_refine(name):
    """prepare_prepass.wgsl + refine_tiles.wgsl + functions.wgsl executed (fixture) vs the oracle's restatement: the same
    final list in the same order, the same indirect args and per-pass tile counts; and the oracle's own derivation of
    the view uniforms still gives the byte-identical struct the fixture was made from"""
    g = np.load(os.path.join(GOLDEN, "refine.npz"))
    total = 0
    for k, view, tiles, indirect, passes in refine_frames(g, name):
        ours, ours_indirect, ours_passes = O.refine(view)
        assert np.array_equal(ours, tiles), (name, k)
        assert ours_indirect == indirect and ours_passes == passes
        total += len(tiles)
    assert total > 1000


def test_oracle_reproduces_golden_formats():
    g = np.load(os.path.join(GOLDEN, "formats.npz"))
    coords = [tuple(int(v) for v in c) for c in g["tc_coords"]]
    assert O.tc_encode(coords) == g["tc_bytes"].tobytes()
    assert [tuple(c) for c in O.tc_decode(g["tc_bytes"].tobytes())] == coords
    assert np.array_equal(np.asarray(O.generate_mipmaps(O.FORMAT_R16, g["mip_source"], 4)), g["mip_chain"])


@pytest.mark.gpu
@pytest.mark.parametrize("generic", [False, True])
@pytest.mark.parametrize("name", CASES)
def test_product_reproduces_golden(name, generic):
    import bevy_terrain_amd as bt

    g = np.load(os.path.join(GOLDEN, name + ".npz"))
    fmt, T, b, lods = (int(v) for v in g["params"])
    cube = name.startswith("cube")
    device = bt.Device(0)
    kw = {} if cube else dict(model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=128, path="terrains/golden", **kw)
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b,
                                           format=bt.AttachmentFormat.R16 if fmt == O.FORMAT_R16 else bt.AttachmentFormat.Rgba8))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    pre = bt.Preprocessor.new().clear_attachment(0, atlas)
    if cube:
        paths = [f"face{s}" for s in range(6)]
        for p, f in zip(paths, g["source"]):
            server.insert(p, np.ascontiguousarray(f))
        pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
    else:
        for k, (src, r) in enumerate(zip(g["source"], g["rects"])):
            server.insert(f"src{k}", np.ascontiguousarray(src))
            pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path=f"src{k}", lod_range=range(0, lods), top_left=(float(r[0]), float(r[1])),
                                                     bottom_right=(float(r[2]), float(r[3]))), server, atlas)
    pre.run(atlas, generic=generic)
    assert [[c.side, c.lod, c.x, c.y, i] for c, i in atlas.tiles()] == g["coords"].tolist()
    data = atlas.download_tiles(0, 0, int(g["coords"][:, 4].max()) + 1)
    check_tiles(g, lambda i: data[i])


@pytest.mark.gpu
@pytest.mark.parametrize("name", REFINE_PATHS)
def test_product_reproduces_golden_refine(name):
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi

    g = np.load(os.path.join(GOLDEN, "refine.npz"))
    prepass = bt.TilingPrepass(bt.Device(0), 100000)
    for k, view, tiles, indirect, passes in refine_frames(g, name):
        assert C_sizeof(_ffi.ViewStateC) == C_sizeof(O.View)  # bt_view_state and orc_view: the same fields in the same order
        v = _ffi.ViewStateC.from_buffer_copy(bytes(view))
        prepass.run(v)
        ours, ours_indirect = prepass.read()
        assert np.array_equal(ours, tiles), (name, k)
        assert list(ours_indirect) == indirect


@pytest.mark.gpu
def test_product_reproduces_golden_formats():
    import bevy_terrain_amd as bt

    g = np.load(os.path.join(GOLDEN, "formats.npz"))
    coords = [bt.TileCoordinate(*(int(v) for v in c)) for c in g["tc_coords"]]
    assert bt.tc_encode(coords) == g["tc_bytes"].tobytes()
    assert [(c.side, c.lod, c.x, c.y) for c in bt.tc_decode(g["tc_bytes"].tobytes())] == [tuple(int(v) for v in c) for c in g["tc_coords"]]
    out = bt.generate_mipmaps(bt.Device(0), bt.AttachmentFormat.R16, g["mip_source"], 4)
    assert np.array_equal(out, g["mip_chain"])
