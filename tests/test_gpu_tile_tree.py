"""TileTree on the GPU (update / adjust_to_tile_atlas / sample_height) + the streaming
TileAtlasState of the library against the oracle's f64 restatement, frame by frame along scripted camera paths."""
import math
import os

import numpy as np
import pytest

import _cases as K
import _oracle as O
import _second_models as S
import bevy_terrain_amd as bt
from test_tile_tree_host import MODELS, positions

pytestmark = pytest.mark.gpu
FUZZ = int(os.environ.get("BT_FUZZ_OFFSET", "0"))  # other seeds of the same sweep: BT_FUZZ_OFFSET=1000 pytest -m gpu -k ...


@pytest.fixture(scope="module")
def device():
    return bt.Device(0)


def camera_path(kind, n, seed=99):
    """a descending spiral towards the terrain (SURVEY §8d refinement input), deterministic"""
    rng = np.random.default_rng(seed)
    out = []
    for i in range(n):
        t = i / max(n - 1, 1)
        a = 2 * math.pi * 2.5 * t + rng.random() * 0.01
        if kind == "planar":
            r = 600.0 * (1 - 0.9 * t)
            out.append((10.0 + r * math.cos(a), 800.0 * (1 - t) + 30.0, 3.0 + r * math.sin(a)))
        else:
            d = np.array([0.4 + 0.9 * (1 - t) * math.cos(a), 0.8, 0.3 + 0.9 * (1 - t) * math.sin(a)])
            d /= np.linalg.norm(d)
            out.append(tuple(d * (6371000.0 + 3.0e6 * (1 - t) ** 2 + 2.0e3)))
    return out


def dummy_atlas(device, model, lod_count):
    cfg = bt.TerrainConfig(lod_count=lod_count, atlas_size=16, path="terrains/none", model=model)
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=16, border_size=2))
    return bt.TileAtlas.new(cfg, device)


@pytest.mark.parametrize("kind", ["planar", "sphere", "ellipsoid"])
@pytest.mark.parametrize("lods,tree_size", [(7, 8), (12, 4)])
def test_update_lists_and_node_tables_equal_the_oracle(device, kind, lods, tree_size):
    model, omodel = MODELS[kind]
    vc = bt.TerrainViewConfig(tree_size=tree_size)
    tree = bt.TileTree(dummy_atlas(device, model, lods), model, lods, vc)
    otree = O.TileTree(omodel, lods, O.make_view_config(tree_size=tree_size))
    total_req = 0
    for frame, pos in enumerate(camera_path(kind, 40)):
        released, requested = tree.update(pos)
        exp_released, exp_requested = otree.update(pos)
        assert released == exp_released, (frame, pos)
        assert requested == exp_requested, (frame, pos)  # the same tiles in the same (push) order
        entries, origins, coords, flags = tree.read()
        e2, o2, c2, f2 = otree.read()
        assert np.array_equal(origins, o2) and np.array_equal(coords, c2) and np.array_equal(flags, f2), frame
        total_req += len(requested)
    assert total_req > 100


def build_terrain(device, tmp_path, model, lod_count, T=32, b=2, seed=3):
    """preprocess a small terrain and save it: the streaming tests load it back tile by tile"""
    W = 2 ** (lod_count - 1) * (T - 2 * b) + 13
    if model.is_spherical():
        cfg = bt.TerrainConfig(lod_count=lod_count, atlas_size=6 * 400, path="terrains/stream", model=model)
    else:
        cfg = bt.TerrainConfig(lod_count=lod_count, atlas_size=400, path="terrains/stream", model=model)
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    pre = bt.Preprocessor.new().clear_attachment(0, atlas)
    if model.is_spherical():
        paths = [f"face{s}" for s in range(6)]
        for s, p in enumerate(paths):
            server.insert(p, K.smooth_raster(W, W, seed=seed + s))
        pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lod_count)), server, atlas)
    else:
        server.insert("src", K.smooth_raster(W, W, seed=seed))
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lod_count)), server, atlas)
    pre.run(atlas)
    root = str(tmp_path / "assets")
    pre.save(atlas, root)
    tiles = {(c.side, c.lod, c.x, c.y): atlas.download_tile(0, i) for c, i in atlas.tiles()}
    return root, cfg, tiles


@pytest.mark.parametrize("kind,atlas_size", [("planar", 256), ("planar", 40), ("sphere", 512)])
def test_streaming_loop_entries_and_heights(device, tmp_path, kind, atlas_size):
    """The reference's per-frame chain (plugin.rs:46-56): compute_requests -> TileAtlas::update -> adjust_to_tile_atlas
    -> approximate_height, product vs oracle in lock step: same request / release lists, same atlas slots (LRU), same
    best-tile table, and sample_height within float tolerance.  atlas_size 40 (at most 1 + 4 + 16 + 16 = 37 tiles are live at once) with a low sweep across the whole terrain forces slot reuse (eviction)."""
    model, omodel = MODELS[kind]
    lods, T, b = 4, 32, 2
    root, cfg, tiles = build_terrain(device, tmp_path, model, lods, T, b)
    stream_cfg = bt.TerrainConfig(lod_count=lods, atlas_size=atlas_size, path=cfg.path, model=model)
    stream_cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16, mip_level_count=3))
    atlas = bt.TileAtlas.new(stream_cfg, device)
    atlas.load_tile_config(root)
    vc = bt.TerrainViewConfig(tree_size=4, load_distance=1.2, blend_distance=1.0)
    ovc = O.make_view_config(tree_size=4, load_distance=1.2, blend_distance=1.0)
    tree = bt.TileTree.new(atlas, vc)
    otree = O.TileTree(omodel, lods, ovc)
    stream = O.Stream(atlas_size, 1, existing=list(tiles))
    smodel = S.StreamModel(atlas_size, 1, existing=list(tiles))  # the second model of the slot state machine, driven by the DEVICE's lists
    layers = {}  # oracle's copy of the atlas contents: atlas_index -> texels
    rng = np.random.default_rng(17)
    loaded_total, evictions = 0, 0
    path = camera_path(kind, 30, seed=5)
    if atlas_size == 40:  # a low pass over the whole terrain: far more distinct tiles than slots
        path = [(10.0 + 450.0 * math.sin(0.37 * i), 40.0, 3.0 + 450.0 * math.sin(0.23 * i + 1.0)) for i in range(60)]
    for frame, pos in enumerate(path):
        # TileTree::compute_requests
        released, requested = tree.update(pos)
        assert (released, requested) == otree.update(pos), frame
        # TileAtlas::update: finish the loads queued by earlier frames, then this frame's releases / requests
        pending = stream.pending_loads()
        assert atlas.pending_loads() == pending
        loaded, failed = atlas.update(root)
        assert (loaded, failed) == (pending, 0)
        finished = stream.finish_loads(pending)
        assert smodel.pending_loads() == pending and smodel.finish_loads(pending) == finished
        for coord, index in finished:
            evictions += index in layers
            layers[index] = tiles[coord]
        for coord in released:  # TileAtlas::update (tile_atlas.rs:590-600): releases, then requests
            smodel.release_tile(coord)
        for coord in requested:
            smodel.request_tile(coord)
        loaded_total += loaded
        tree.apply_requests()
        otree.apply_requests(stream)
        # TileTree::adjust_to_tile_atlas
        tree.adjust_to_tile_atlas()
        otree.adjust_to_tile_atlas(stream)
        entries, origins, coords, flags = tree.read()
        e2, o2, c2, f2 = otree.read()
        assert np.array_equal(coords, c2) and np.array_equal(entries, e2), frame
        # adjust_to_tile_atlas (tile_tree.rs:337-386) once more from the slot model: every node's entry is the best loaded tile of its coordinate
        m_entries = np.array([smodel.get_best_tile(tuple(int(v) for v in c)) if c[1] != 0xFFFFFFFF else (0xFFFFFFFF, 0xFFFFFFFF) for c in coords], np.uint32)
        assert np.array_equal(entries, m_entries), frame
        # every loaded slot holds the bytes of its tile
        for coord in list(tiles)[:: max(1, len(tiles) // 7)]:
            idx, lod = atlas.get_best_tile(bt.TileCoordinate(*coord))
            assert (idx, lod) == stream.get_best_tile(coord) == smodel.get_best_tile(coord)
            if lod == coord[1]:
                assert np.array_equal(atlas.download_tile(0, idx), tiles[coord])
        # TileTree::approximate_height + a batch of sample_height queries around the view
        h_before = tree.view_state().approximate_height  # what surface_position uses while the new height is sampled
        h = tree.approximate_height()
        _, exp_h = otree.sample_attachment(O.FORMAT_R16, T, b, layers, [pos])
        assert h == float(exp_h[0]), (frame, h, float(exp_h[0]))
        otree.set_approximate_height(h)  # both sides continue from the same (f32) value
        if kind == "planar":
            pts = np.column_stack([rng.uniform(-480, 480, 64) + 10.0, rng.uniform(0, 300, 64), rng.uniform(-480, 480, 64) + 3.0])
        else:
            pts = np.asarray(pos) + rng.normal(size=(64, 3)) * 4.0e5
        ours, ours_h = tree.sample_attachment(0, pts)
        exp, exp_heights = otree.sample_attachment(O.FORMAT_R16, T, b, layers, pts)
        # compute_blend's f64 log2 is OCML's on the device and libm's in the oracle (<= 1 ULP of a double apart): its conversion
        # to f32 absorbs that except within 2^-29 of a rounding boundary — with these fixed seeds every value is bit-equal
        assert np.array_equal(ours, exp), (frame, np.abs(ours - exp).max(), int((ours != exp).sum()))
        assert np.array_equal(ours_h, exp_heights), (frame, np.abs(ours_h - exp_heights).max())
        # ... and a second, independent model of sample_attachment (tests/_second_models.py: numpy, one sample at a time, from the
        # reference text; it takes the best-tile table and the loaded texels as given)
        sides = 6 if kind != "planar" else 1
        model_scale = float(model.scale_vec[0]) / 2.0 if kind == "planar" else float(model.scale_vec[0])
        mine = S.TileTreeModel(kind, model.translation, model.scale_vec, model.min_height, model.max_height, lods, 4, 1.2)
        m_values, m_heights = S.sample_attachment_r16(mine, pos, h, 1.0 * model_scale, vc.blend_range, lods, entries.reshape(sides, lods, 4, 4, 2),
                                                      layers, T, b, pts)
        assert np.array_equal(ours, m_values), (frame, np.abs(ours - m_values).max(), int((ours != m_values).sum()))
        # (TileTree::approximate_height, tile_tree.rs:376-386: sample_height at the view position, with the previous height)
        assert S.sample_attachment_r16(mine, pos, h_before, 1.0 * model_scale, vc.blend_range, lods, entries.reshape(sides, lods, 4, 4, 2), layers, T, b, [pos])[1][0] == h
        assert np.array_equal(ours_h, m_heights), (frame, np.abs(ours_h - m_heights).max())
    assert loaded_total > 20
    if atlas_size == 40:
        assert evictions > 0
    # the prepass input derived from the tree's state == bt_view_state_from_config of the same state
    v = tree.view_state()
    assert bytes(v) == bytes(bt.view_state_from_config(model, vc, path[-1], v.approximate_height))


@pytest.mark.parametrize("kind,unordered", [("planar", False), ("sphere", False), ("sphere", True)])
def test_frame_update_equals_the_separate_calls(device, tmp_path, kind, unordered):
    """bt_frame_update (one call, one host synchronisation, the height left on the device) against the chain of separate calls
    it replaces — two streaming instances of the same terrain in lock step: the same request / release lists, the same
    tile-tree entries, the same final tile list of the prepass (ordered form: in order; unordered form: as a set) and the
    same indirect arguments, frame by frame; the host's copy of the height arrives one frame late and is then the same."""
    model, _ = MODELS[kind]
    lods, T, b = 4, 32, 2
    root, cfg, tiles = build_terrain(device, tmp_path, model, lods, T, b)
    vc = bt.TerrainViewConfig(tree_size=4, load_distance=1.2, blend_distance=1.0, geometry_tile_count=20000)

    def instance():
        scfg = bt.TerrainConfig(lod_count=lods, atlas_size=256 if kind == "planar" else 512, path=cfg.path, model=model)
        scfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16, mip_level_count=3))
        atlas = bt.TileAtlas.new(scfg, device)
        atlas.load_tile_config(root)
        return atlas, bt.TileTree.new(atlas, vc), bt.TilingPrepass(device, vc.geometry_tile_count)

    atlas_a, tree_a, prepass_a = instance()
    atlas_b, tree_b, prepass_b = instance()
    heights_a, moved = [], 0
    for frame, pos in enumerate(camera_path(kind, 25, seed=5)):
        # A: the separate calls, in the reference's order (plugin.rs:46-56)
        released, requested = tree_a.update(pos)
        atlas_a.update(root)
        tree_a.apply_requests()
        tree_a.adjust_to_tile_atlas()
        heights_a.append(tree_a.approximate_height())
        prepass_a.run(tree_a.view_state(), unordered=unordered)
        # B: the loads of earlier frames, then ONE call
        atlas_b.update(root)
        info = tree_b.frame_update(pos, prepass_b, unordered=unordered)
        assert (info.released_count, info.requested_count, info.apply_status) == (len(released), len(requested), 0), frame
        moved += len(requested)
        if frame > 0:  # the height the update of this frame used = the sample of the previous frame
            assert info.approximate_height == heights_a[frame - 1], frame
        ea, oa, ca, fa = tree_a.read()
        eb, ob, cb, fb = tree_b.read()
        assert np.array_equal(ea, eb) and np.array_equal(oa, ob) and np.array_equal(ca, cb) and np.array_equal(fa, fb), frame
        ta, ia = prepass_a.read()
        tb, ib = prepass_b.read()
        assert list(ia) == list(ib), frame
        if unordered:
            assert sorted(map(tuple, ta)) == sorted(map(tuple, tb)), frame
        else:
            assert np.array_equal(ta, tb), frame
        assert len(ta) > 0
        # after the synchronisation of read(): the host's copy has caught up
        assert tree_b.view_state().approximate_height == heights_a[frame], frame
    assert moved > 20


def test_mips_of_streamed_tiles(device, tmp_path):
    model, _ = MODELS["planar"]
    root, cfg, tiles = build_terrain(device, tmp_path, model, 3)
    scfg = bt.TerrainConfig(lod_count=3, atlas_size=64, path=cfg.path, model=model)
    scfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=32, border_size=2, format=bt.AttachmentFormat.R16, mip_level_count=3))
    atlas = bt.TileAtlas.new(scfg, device)
    atlas.load_tile_config(root)
    for c in [(0, 0, 0, 0), (0, 2, 3, 1), (0, 1, 1, 1)]:
        atlas.request_tile(bt.TileCoordinate(*c))
    assert atlas.pending_loads() == 3 and atlas.get_best_tile(bt.TileCoordinate(0, 2, 3, 1)) == (O.INVALID, O.INVALID)
    assert atlas.update(root, max_loads=2) == (2, 0) and atlas.pending_loads() == 1
    assert atlas.get_best_tile(bt.TileCoordinate(0, 2, 3, 1))[1] == 2 and atlas.get_best_tile(bt.TileCoordinate(0, 1, 1, 1)) == (0, 0)
    assert atlas.update(root) == (1, 0)
    for c in [(0, 0, 0, 0), (0, 2, 3, 1), (0, 1, 1, 1)]:
        idx, lod = atlas.get_best_tile(bt.TileCoordinate(*c))
        assert lod == c[1]
        chain = O.generate_mipmaps(O.FORMAT_R16, tiles[c], 3)
        assert np.array_equal(atlas.download_mip(0, 1, idx).ravel(), chain[32 * 32:32 * 32 + 16 * 16])
        assert np.array_equal(atlas.download_mip(0, 2, idx).ravel(), chain[32 * 32 + 16 * 16:])
    # a missing file: the tile stays Loading forever, like the reference (tile_atlas.rs:202-204)
    os.remove(os.path.join(root, cfg.path, "data/height/0_2_0_0.bin"))
    atlas.request_tile(bt.TileCoordinate(0, 2, 0, 0))
    assert atlas.update(root) == (0, 1)
    assert atlas.get_best_tile(bt.TileCoordinate(0, 2, 0, 0)) == (0, 0)
    with pytest.raises(bt._ffi.BtError):
        atlas.release_tile(bt.TileCoordinate(0, 2, 1, 1))  # existing, but never requested


def test_tile_is_loaded_only_when_all_its_attachments_are(device, tmp_path):
    """LoadingState::Loading(n) (tile_atlas.rs:347-359): a tile with two attachments becomes usable after BOTH loads."""
    model, _ = MODELS["planar"]
    T, lods = 16, 2
    src_h = K.random_raster(O.FORMAT_R16, 40, 40, seed=1)
    src_a = K.random_raster(O.FORMAT_RGBA8, 40, 40, seed=2)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=16, path="terrains/two", model=model)
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=2, format=bt.AttachmentFormat.R16))
    cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=T, border_size=2, format=bt.AttachmentFormat.Rgba8))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("h", src_h).insert("a", src_a)
    pre = (bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, lods)), server, atlas)
           .preprocess_tile(bt.PreprocessDataset(attachment_index=1, path="a", lod_range=range(0, lods)), server, atlas))
    pre.run(atlas)
    root = str(tmp_path / "assets")
    pre.save(atlas, root)
    originals = {(c.side, c.lod, c.x, c.y): (atlas.download_tile(0, i), atlas.download_tile(1, i)) for c, i in atlas.tiles()}
    fresh = bt.TileAtlas.new(cfg, device)
    fresh.load_tile_config(root)
    stream = O.Stream(16, 2, existing=list(originals))
    for c in [(0, 0, 0, 0), (0, 1, 1, 0)]:
        fresh.request_tile(bt.TileCoordinate(*c))
        stream.request_tile(c)
    assert fresh.pending_loads() == stream.pending_loads() == 4
    for k in range(4):  # one attachment at a time
        assert fresh.update(root, max_loads=1) == (1, 0)
        stream.finish_loads(1)
        for c in [(0, 0, 0, 0), (0, 1, 1, 0), (0, 1, 0, 1)]:
            assert fresh.get_best_tile(bt.TileCoordinate(*c)) == stream.get_best_tile(c), (k, c)
    idx, lod = fresh.get_best_tile(bt.TileCoordinate(0, 1, 1, 0))
    assert lod == 1
    assert np.array_equal(fresh.download_tile(0, idx), originals[(0, 1, 1, 0)][0])
    assert np.array_equal(fresh.download_tile(1, idx), originals[(0, 1, 1, 0)][1])


def draw_tree_case(seed):
    rng = np.random.default_rng(77_000 + seed)
    kind = ["planar", "sphere", "ellipsoid"][seed % 3]
    centre = tuple(float(v) for v in rng.uniform(-500.0, 500.0, 3))
    if kind == "planar":
        side = float(rng.choice([10.0, 1000.0, 250000.0]))
        lo, hi = 0.0, float(rng.choice([1.0, 0.25 * side]))
        model, omodel = bt.TerrainModel.planar(centre, side, lo, hi), O.make_model("planar", centre, side, 0.0, lo, hi)
        scale = side
    elif kind == "sphere":
        radius = float(rng.choice([50.0, 6371000.0]))
        lo, hi = -0.002 * radius, 0.0015 * radius
        model, omodel = bt.TerrainModel.sphere(centre, radius, lo, hi), O.make_model("spherical", centre, radius, 0.0, lo, hi)
        scale = radius
    else:
        major = float(rng.choice([100.0, 6378137.0]))
        minor = major * float(rng.choice([0.5, 0.9966, 1.0]))
        lo, hi = -0.002 * major, 0.0015 * major
        model, omodel = bt.TerrainModel.ellipsoid(centre, major, minor, lo, hi), O.make_model("ellipsoidal", centre, major, minor, lo, hi)
        scale = major
    lods = int(rng.integers(1, 16))
    cfg = dict(tree_size=int(rng.choice([2, 4, 8, 16])), load_distance=float(rng.choice([0.6, 2.5, 5.0])),
               blend_distance=float(rng.choice([1.0, 2.0])), origin_lod=int(rng.integers(0, 14)))
    pts = []
    for _ in range(24):  # teleports, not a smooth path: far away, skimming the surface, inside the body, on face edges
        u = rng.random()
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if kind == "planar":
            p = np.array(centre) + np.array([rng.uniform(-0.8, 0.8) * scale, rng.choice([1e-3, 0.01, 0.3, 3.0]) * scale, rng.uniform(-0.8, 0.8) * scale])
        elif u < 0.2:  # exactly above a cube edge / corner / face centre
            e = np.array(rng.choice([-1.0, 0.0, 1.0], 3))
            e = e if np.any(e) else np.array([0.0, 1.0, 0.0])
            p = np.array(centre) + e / np.linalg.norm(e) * scale * rng.choice([1.0001, 1.3, 8.0])
        else:
            p = np.array(centre) + d * scale * rng.choice([0.4, 0.999, 1.00001, 1.01, 2.0, 50.0])
        pts.append(tuple(float(v) for v in p))
    return model, omodel, lods, cfg, pts


@pytest.mark.parametrize("seed", range(FUZZ, FUZZ + 45))
def test_update_random_models_configs_and_teleports(device, seed):
    model, omodel, lods, cfg, pts = draw_tree_case(seed)
    vc = bt.TerrainViewConfig(**cfg)
    tree = bt.TileTree(dummy_atlas(device, model, lods), model, lods, vc)
    otree = O.TileTree(omodel, lods, O.make_view_config(**cfg))
    for frame, pos in enumerate(pts):
        released, requested = tree.update(pos)
        exp_released, exp_requested = otree.update(pos)
        assert released == exp_released, (seed, frame, pos)
        assert requested == exp_requested, (seed, frame, pos)
        entries, origins, coords, flags = tree.read()
        e2, o2, c2, f2 = otree.read()
        assert np.array_equal(origins, o2) and np.array_equal(coords, c2) and np.array_equal(flags, f2), (seed, frame)
    tree.close()
