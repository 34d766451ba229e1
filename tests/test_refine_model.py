"""The tiling prepass' two independent CPU statements agree: oracle/bt_oracle.c (sequential, statement by statement)
and tests/_refine_model.py (whole passes as numpy float32 arrays) produce the same final tile LIST, and the result
satisfies the quadtree invariants SURVEY §4 item 4 names (disjoint, covers all roots, neighbour LOD difference)."""
import numpy as np
import pytest

import _oracle as O
import _refine_model as R
import bevy_terrain_amd as bt
from test_tile_tree_host import MODELS, positions


def frames(kind, n):
    model, _ = MODELS[kind]
    for p in positions(kind, n, 7):
        if kind != "planar":
            p = p / np.linalg.norm(p) * (6371000.0 + 10 ** np.random.default_rng(int(abs(p[0])) % 1000).uniform(2.5, 6.5))
        yield model, tuple(p)


def oracle_view(v):
    return O.make_view(spherical=v.spherical, tile_count=v.geometry_tile_count, refinement_count=v.refinement_count,
                       vertices_per_tile=v.vertices_per_tile, subdivision_distance=v.subdivision_distance,
                       origin_lod=v.origin_lod, approximate_height=v.approximate_height,
                       sides=[((s.view_xy[0], s.view_xy[1]), (s.view_uv[0], s.view_uv[1])) for s in v.sides],
                       world_position=list(v.world_position), world_from_local=list(v.world_from_local),
                       local_from_world_transpose=list(v.local_from_world_transpose))


@pytest.mark.parametrize("kind", ["planar", "sphere"])
def test_numpy_model_equals_the_sequential_oracle(kind):
    cfg = bt.TerrainViewConfig(geometry_tile_count=400000)
    worst_seen, total = 0, 0
    for model, pos in frames(kind, 12):
        v = bt.make_view_state(model, cfg, pos)
        exp, indirect, passes = O.refine(oracle_view(v))
        final, dropped, counts = R.refine(v)
        assert np.array_equal(final, exp), pos
        assert counts[:len(passes)] == passes[:len(counts)]
        assert indirect[0] == v.vertices_per_tile * len(final)
        assert len(dropped) == 0  # the default 30 passes are never exhausted here
        worst = R.check_quadtree(final, dropped, 6 if kind == "sphere" else 1)
        worst_seen = max(worst_seen, worst)
        total += len(final)
    assert total > 1000
    # distance-based subdivision with the default morph_distance 16 / tolerance 0.1: edge-adjacent tiles of one face differ
    # by at most one LOD (what the vertex morph needs, docs/implementation.md)
    assert worst_seen <= 1


def test_depth_limit_drops_dividing_tiles_and_the_cover_accounts_for_them():
    model, _ = MODELS["planar"]
    cfg = bt.TerrainViewConfig(geometry_tile_count=100000, refinement_count=3, morph_distance=1.0)
    v = bt.make_view_state(model, cfg, (20.0, 126.0, 10.0))  # just above the approximate surface (height 125)
    exp, _, passes = O.refine(oracle_view(v))
    final, dropped, counts = R.refine(v)
    assert np.array_equal(final, exp) and counts == passes
    assert len(dropped) > 0 and dropped[:, 1].min() == 3
    R.check_quadtree(final, dropped, 1)  # final + dropped tiles tile the root exactly


def test_should_be_divided_matches_the_oracle_per_tile():
    model, _ = MODELS["sphere"]
    v = bt.make_view_state(model, bt.TerrainViewConfig(), (6.0e6, 3.0e6, -1.5e6))
    ov = oracle_view(v)
    rng = np.random.default_rng(4)
    lod = rng.integers(0, 14, 3000)
    tiles = np.stack([rng.integers(0, 6, 3000), lod, rng.integers(0, 1 << 14, 3000) >> (14 - lod), rng.integers(0, 1 << 14, 3000) >> (14 - lod)], axis=1).astype(np.uint32)
    divide, dist = R.should_be_divided(v, tiles)
    for t, dv, ds in zip(tiles.tolist(), divide.tolist(), dist.tolist()):
        exp_div, exp_dist = O.should_be_divided(ov, tuple(t))
        assert (dv, np.float32(ds)) == (exp_div, np.float32(exp_dist)), t
