"""SURVEY §8 rows a20 / f1 (host side): the view-state derivation of the library (bt_view_state_from_config, f64 on
the host, no GPU involved) against the oracle's restatement and against hand-computed values; known answers and
invariants of the oracle's TileTree / streaming TileAtlasState / ellipsoid projection restatements."""
import ctypes as C
import math

import numpy as np
import pytest

import _oracle as O
import bevy_terrain_amd as bt
from bevy_terrain_amd import _ffi
from bevy_terrain_amd.tile_tree import model_c, view_config_c

MODELS = {
    "planar": (bt.TerrainModel.planar((10.0, -5.0, 3.0), 1000.0, 0.0, 250.0), O.make_model("planar", (10.0, -5.0, 3.0), 1000.0, 0.0, 0.0, 250.0)),
    "sphere": (bt.TerrainModel.sphere((0.0, 0.0, 0.0), 6371000.0, -12000.0, 9000.0), O.make_model("spherical", (0, 0, 0), 6371000.0, 0.0, -12000.0, 9000.0)),
    "ellipsoid": (bt.TerrainModel.ellipsoid((100.0, 200.0, -300.0), 6378137.0, 6356752.314245, -12000.0, 9000.0),
                  O.make_model("ellipsoidal", (100.0, 200.0, -300.0), 6378137.0, 6356752.314245, -12000.0, 9000.0)),
}


def struct_bytes(s):
    return bytes(C.string_at(C.addressof(s), C.sizeof(s)))


def positions(kind, n, seed):
    rng = np.random.default_rng(seed)
    if kind == "planar":
        p = rng.uniform(-700.0, 700.0, size=(n, 3))
        p[:, 1] = rng.uniform(1.0, 900.0, size=n)
        return p
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return d * (6371000.0 + rng.uniform(100.0, 5.0e6, size=(n, 1)))


@pytest.mark.parametrize("kind", ["planar", "sphere", "ellipsoid"])
def test_view_state_from_config_equals_the_oracle_restatement(kind):
    model, omodel = MODELS[kind]
    vc = bt.TerrainViewConfig(geometry_tile_count=123456, refinement_count=17, grid_size=12, origin_lod=9)
    ovc = O.make_view_config(geometry_tile_count=123456, refinement_count=17, grid_size=12, origin_lod=9)
    assert C.sizeof(_ffi.ViewStateC) == C.sizeof(O.View)
    pts = list(positions(kind, 200, 5))
    # axis-aligned and face-edge positions: every branch of Coordinate::from_world_position
    if kind != "planar":
        r = 7.0e6
        pts += [(r, 0, 0), (-r, 0, 0), (0, r, 0), (0, -r, 0), (0, 0, r), (0, 0, -r), (r, r, 0.5 * r), (-r, 0.3 * r, -r), (r, r, r)]
    for p in pts:
        ours = bt.view_state_from_config(model, vc, tuple(p), 321.5)
        theirs = O.view_state_from_config(omodel, ovc, tuple(p), 321.5)
        assert struct_bytes(ours) == struct_bytes(theirs), p


def test_view_state_known_answers_planar():
    # examples/minimal.rs: side 1000, heights 0..250, default view config: hand-computed
    model = bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 250.0)
    v = bt.view_state_from_config(model, bt.TerrainViewConfig(), (100.0, 300.0, -200.0), 125.0)
    assert (v.spherical, v.geometry_tile_count, v.refinement_count, v.vertices_per_tile, v.origin_lod) == (0, 1000000, 30, 2 * 16 * 18, 10)
    assert v.subdivision_distance == np.float32(16.0 * 500.0 * 1.1)
    assert v.approximate_height == 125.0
    # uv = (100/1000 + 0.5, -200/1000 + 0.5) = (0.6, 0.3); x 1024 = 614.4, 307.2
    assert (v.sides[0].view_xy[0], v.sides[0].view_xy[1]) == (614, 307)
    assert v.sides[0].view_uv[0] == np.float32(0.6 * 1024 - 614) and v.sides[0].view_uv[1] == np.float32(0.3 * 1024 - 307)
    assert list(v.world_position) == [100.0, 300.0, -200.0]
    assert list(v.world_from_local) == [1000.0, 0, 0, 0, 1000.0, 0, 0, 0, 1000.0, 0, 0, 0]
    assert list(v.local_from_world_transpose) == [np.float32(1e-3), 0, 0, 0, np.float32(1e-3), 0, 0, 0, np.float32(1e-3)]
    # the Python host mirror used by the other tests derives the same struct
    assert struct_bytes(v) == struct_bytes(bt.make_view_state(model, bt.TerrainViewConfig(), (100.0, 300.0, -200.0)))


def test_view_state_sphere_sides_follow_the_projection_tables():
    sphere, _ = MODELS["sphere"]
    v = bt.view_state_from_config(sphere, bt.TerrainViewConfig(), (7e6, 1e5, -2e5), 0.0)
    assert v.spherical == 1
    # the view is above the +x face (side 3), close to its centre
    assert abs(v.sides[3].view_xy[0] - 512) < 32 and abs(v.sides[3].view_xy[1] - 512) < 32
    # every other side holds the closest point of that face: one coordinate pinned to an edge (0 or 1024), the
    # opposite face (side 0) both taken over (coordinate.rs:27-42)
    for side in (1, 2, 4, 5):
        xy = (v.sides[side].view_xy[0], v.sides[side].view_xy[1])
        assert xy[0] in (0, 1024) or xy[1] in (0, 1024), (side, xy)
    assert struct_bytes(v) == struct_bytes(bt.make_view_state(sphere, bt.TerrainViewConfig(), (7e6, 1e5, -2e5), approximate_height=0.0))


def test_coordinate_round_trip_and_face_selection():
    _, sphere = MODELS["sphere"]
    for p in positions("sphere", 100, 11):
        side, uv = O.coordinate_from_world_position(sphere, p)
        assert 0 <= side < 6 and 0.0 <= uv[0] <= 1.0 and 0.0 <= uv[1] <= 1.0
        back = np.array(O.coordinate_world_position(sphere, side, uv, 0.0))
        assert np.allclose(back, p / np.linalg.norm(p) * 6371000.0, rtol=0, atol=1e-6)  # the point below the view, on the sphere
    _, planar = MODELS["planar"]
    side, uv = O.coordinate_from_world_position(planar, (10.0 + 250.0, 77.0, 3.0 - 500.0))
    assert side == 0 and uv == (0.75, 0.0)
    assert O.coordinate_from_world_position(planar, (1e9, 0.0, -1e9))[1] == (1.0, 0.0)  # clamped to the terrain


def test_project_point_ellipsoid_known_answers():
    a, b = 6378137.0, 6356752.314245
    # the call site passes e = (major, major, minor) (terrain_model.rs:163-171) and the function works on y.xzy()
    # (ellipsoid.rs:14): the minor (polar) axis is the world's Y, like the model's scale (major, minor, major)
    e = (a, a, b)
    assert O.project_point_ellipsoid(e, (a, 0.0, 0.0)) == pytest.approx((a, 0.0, 0.0))
    assert O.project_point_ellipsoid(e, (0.0, 0.0, 2 * b)) == pytest.approx((0.0, 0.0, a))
    assert O.project_point_ellipsoid(e, (0.0, -3 * a, 0.0)) == pytest.approx((0.0, -b, 0.0))
    rng = np.random.default_rng(3)
    for _ in range(50):
        y = rng.normal(size=3) * rng.uniform(0.2, 3.0) * a
        x = np.array(O.project_point_ellipsoid(e, y))
        assert (x[0] / a) ** 2 + (x[1] / b) ** 2 + (x[2] / a) ** 2 == pytest.approx(1.0, abs=1e-9)  # on the ellipsoid
        assert np.all(np.sign(x) == np.sign(y))
        # closest point: y - x is parallel to the surface normal (gradient) at x
        n = np.array([x[0] / a ** 2, x[1] / b ** 2, x[2] / a ** 2])
        d = y - x
        cosang = abs(np.dot(n, d)) / (np.linalg.norm(n) * np.linalg.norm(d))
        assert cosang == pytest.approx(1.0, abs=1e-9)


def test_streaming_state_machine_known_answers():
    # tile_atlas.rs:383-503 on a 3-slot atlas with 2 attachments, by hand
    a, b, c, d = (0, 1, 0, 0), (0, 1, 1, 0), (0, 1, 0, 1), (0, 1, 1, 1)
    root = (0, 0, 0, 0)
    s = O.Stream(3, 2, existing=[root, a, b, c, d])
    assert s.request_tile((0, 5, 1, 1)) == 0 and s.pending_loads() == 0  # not an existing tile: ignored (:419-421)
    assert s.request_tile(root) == 0 and s.atlas_index(root) == 0 and s.pending_loads() == 2
    assert s.get_best_tile(a) == (O.INVALID, O.INVALID)  # nothing loaded yet, not even the ancestor
    assert s.finish_loads(1) == [(root, 0)] and s.get_best_tile(a) == (O.INVALID, O.INVALID)  # Loading(2) -> Loading(1)
    assert s.finish_loads(1) == [(root, 0)] and s.get_best_tile(a) == (0, 0)  # Loaded: `a` falls back to its parent
    assert s.request_tile(a) == 0 and s.request_tile(b) == 0 and (s.atlas_index(a), s.atlas_index(b)) == (1, 2)
    assert s.request_tile(c) == -2  # "Atlas out of indices"
    s.finish_loads(4)
    assert s.get_best_tile(a) == (1, 1) and s.get_best_tile((0, 2, 3, 1)) == (2, 1)  # grandchild of b -> b
    assert s.request_tile(a) == 0 and s.release_tile(a) == 0 and s.get_best_tile(a) == (1, 1)  # two requests, one release
    assert s.release_tile(a) == 0  # now unused: cached at the back of the LRU ...
    assert s.get_best_tile(a) == (1, 1)  # ... still served
    assert s.release_tile(a) == -1  # "Tried releasing a tile, which is not present."
    assert s.request_tile(c) == 0 and s.atlas_index(c) == 1 and s.atlas_index(a) == O.INVALID  # c evicts a's slot
    assert s.get_best_tile(a) == (0, 0)  # a is gone: back to the root
    s.finish_loads(2)
    assert s.release_tile(b) == 0 and s.request_tile(b) == 0 and s.atlas_index(b) == 2  # re-requested before eviction: same slot
    assert s.request_tile(d) == -2  # and its slot is no longer in the LRU


@pytest.mark.parametrize("kind", ["planar", "sphere", "ellipsoid"])
def test_tile_tree_update_invariants(kind):
    _, omodel = MODELS[kind]
    lods, ts = 6, 8
    tree = O.TileTree(omodel, lods, O.make_view_config(tree_size=ts))
    p0 = positions(kind, 1, 21)[0]
    released, requested = tree.update(p0)
    assert released == [] and len(set(requested)) == len(requested)
    entries, origins, coords, flags = tree.read()
    sides = 1 if kind == "planar" else 6
    coords = coords.reshape(sides, lods, ts, ts, 4)
    flags = flags.reshape(sides, lods, ts, ts)
    for side in range(sides):
        for lod in range(lods):
            n = 1 << lod
            ox, oy = origins[side, lod]
            assert ox + min(ts, n) <= max(n, ts) and oy + min(ts, n) <= max(n, ts)
            # slot (x % ts, y % ts) holds tile (x, y) of the window origin .. origin + ts
            for x in range(ts):
                for y in range(ts):
                    c = coords[side, lod, (ox + x) % ts, (oy + y) % ts]
                    assert tuple(c) == (side, lod, ox + x, oy + y)
            if lod == 0:
                assert flags[side, lod].all()  # lod 0 is always requested (:296)
    assert sorted(requested) == sorted(tuple(c) for c in coords.reshape(-1, 4)[flags.reshape(-1) == 1])
    # the same view again: nothing changes
    assert tree.update(p0) == ([], [])
    # a view on the other side of the terrain: everything that was requested and is no longer wanted is released once
    released, requested2 = tree.update(-p0 if kind != "planar" else (-p0[0], p0[1], -p0[2]))
    assert len(set(released)) == len(released) and set(released) <= set(requested)
    assert not (set(requested2) & (set(requested) - set(released)))


def test_compute_blend_known_answers():
    _, planar = MODELS["planar"]
    tree = O.TileTree(planar, 5, O.make_view_config())  # blend_distance = 2 * 500 = 1000, blend_range 0.2
    tree.update((10.0, 0.0, 3.0))
    # distance 1000 / 2^k -> target_lod = k exactly: ratio = inverse_mix(k + 0.2, k, k) = 1 (k > 0), lod 0 -> 0
    assert tree.compute_blend((10.0 + 1000.0, 0.0, 3.0)) == (0, 0.0)
    assert tree.compute_blend((10.0 + 250.0, 0.0, 3.0)) == (2, 1.0)
    lod, ratio = tree.compute_blend((10.0 + 1000.0 / 2 ** 2.1, 0.0, 3.0))  # target 2.1: halfway through the blend range
    assert lod == 2 and ratio == pytest.approx(0.5, abs=1e-5)
    assert tree.compute_blend((10.0 + 1000.0 / 2 ** 2.5, 0.0, 3.0)) == (2, 0.0)
    assert tree.compute_blend((10.0 + 1e-9, 0.0, 3.0))[0] == 4  # capped at lod_count - 0.00001


def test_null_and_invalid_arguments_are_status_codes():
    L = _ffi.lib()
    v = _ffi.ViewStateC()
    pos = (C.c_double * 3)(0.0, 1.0, 0.0)
    model = model_c(bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    vc = view_config_c(bt.TerrainViewConfig())
    assert L.bt_view_state_from_config(None, C.byref(vc), pos, 0.0, C.byref(v)) == -1
    assert L.bt_view_state_from_config(C.byref(model), None, pos, 0.0, C.byref(v)) == -1
    assert L.bt_view_state_from_config(C.byref(model), C.byref(vc), pos, 0.0, None) == -1
    bad = model_c(bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    bad.kind = 7
    assert L.bt_view_state_from_config(C.byref(bad), C.byref(vc), pos, 0.0, C.byref(v)) == -1 and b"model" in L.bt_last_error()
    bad.kind, bad.a = 2, 1.0  # ellipsoid without a minor axis
    assert L.bt_view_state_from_config(C.byref(bad), C.byref(vc), pos, 0.0, C.byref(v)) == -1
    assert L.bt_tile_tree_create(None, C.byref(model), 4, C.byref(vc), C.byref(C.c_void_p())) == -1
    assert L.bt_atlas_request_tile(None, _ffi.TileCoordinateC(0, 0, 0, 0)) == -1
    assert L.bt_image_decode(None, 0, 1, C.byref(_ffi.ImageC())) == -1
    d = _ffi.TerrainViewConfigC()
    L.bt_terrain_view_config_default(C.byref(d))
    assert (d.tree_size, d.geometry_tile_count, d.refinement_count, d.grid_size, d.origin_lod) == (8, 1000000, 30, 16, 10)
    assert (d.subdivision_tolerance, d.load_distance, d.morph_distance, d.blend_distance) == (0.1, 2.5, 16.0, 2.0)
