"""TileTree (src/terrain_data/tile_tree.rs:103-387) + GpuTileTree (gpu_tile_tree.rs:22-95) behind the reference's
names: `TileTree.new(tile_atlas, view_config)`, `compute_requests` / `update`, `adjust_to_tile_atlas`,
`approximate_height`, and the free functions `sample_attachment` / `sample_height` (terrain_data/mod.rs:265-307).
The node tables live on the GPU; this module is a thin ctypes mirror used by the tests."""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence, Tuple

import numpy as np

from . import _ffi
from .terrain import TerrainModel, TerrainViewConfig, TileCoordinate
from .tile_atlas import TileAtlas, device_open


def model_c(model: TerrainModel) -> _ffi.TerrainModelC:
    m = _ffi.TerrainModelC()
    m.kind = {"planar": 0, "spherical": 1, "ellipsoidal": 2}[model.kind]
    for i in range(3):
        m.position[i] = model.translation[i]
    if model.kind == "planar":
        m.a = model.side_length
    elif model.kind == "spherical":
        m.a = model.radius
    else:
        m.a, m.b = model.major_axis, model.minor_axis
    m.min_height, m.max_height = model.min_height, model.max_height
    return m


def view_config_c(vc: TerrainViewConfig) -> _ffi.TerrainViewConfigC:
    c = _ffi.TerrainViewConfigC()
    for name in ("tree_size", "geometry_tile_count", "refinement_count", "grid_size", "subdivision_tolerance",
                 "precision_threshold_distance", "load_distance", "morph_distance", "blend_distance", "morph_range",
                 "blend_range", "origin_lod"):
        setattr(c, name, getattr(vc, name))
    return c


def view_state_from_config(model: TerrainModel, view_config: TerrainViewConfig, view_world_position: Sequence[float],
                           approximate_height: float) -> _ffi.ViewStateC:
    """bt_view_state_from_config: the prepass' per-frame inputs derived in the library (f64, `as f32` casts)."""
    v = _ffi.ViewStateC()
    pos = (C.c_double * 3)(*view_world_position)
    _ffi.check(_ffi.lib().bt_view_state_from_config(C.byref(model_c(model)), C.byref(view_config_c(view_config)), pos,
                                                    C.c_float(approximate_height), C.byref(v)))
    return v


class TileTree:
    def __init__(self, tile_atlas: TileAtlas, model: TerrainModel, lod_count: int, view_config: TerrainViewConfig):
        self.atlas = tile_atlas
        self.model = model
        self.lod_count = lod_count
        self.view_config = view_config
        self.sides = model.side_count()
        self.nodes = self.sides * lod_count * view_config.tree_size ** 2
        h = C.c_void_p()
        _ffi.check(_ffi.lib().bt_tile_tree_create(tile_atlas.device._h, C.byref(model_c(model)), lod_count,
                                                  C.byref(view_config_c(view_config)), C.byref(h)))
        self._h = h

    @staticmethod
    def new(tile_atlas: TileAtlas, view_config: TerrainViewConfig) -> "TileTree":
        return TileTree(tile_atlas, tile_atlas.config.model, tile_atlas.config.lod_count, view_config)

    def update(self, view_position: Sequence[float]) -> Tuple[List[tuple], List[tuple]]:
        """TileTree::update: returns (released_tiles, requested_tiles) in the reference's push order."""
        pos = (C.c_double * 3)(*view_position)
        _ffi.check(_ffi.lib().bt_tile_tree_update(self._h, pos))
        rel, req = C.POINTER(_ffi.TileCoordinateC)(), C.POINTER(_ffi.TileCoordinateC)()
        nrel, nreq = C.c_uint32(), C.c_uint32()
        _ffi.check(_ffi.lib().bt_tile_tree_requests(self._h, C.byref(rel), C.byref(nrel), C.byref(req), C.byref(nreq)))
        t = lambda c: (c.side, c.lod, c.x, c.y)
        return [t(rel[i]) for i in range(nrel.value)], [t(req[i]) for i in range(nreq.value)]

    def apply_requests(self):
        _ffi.check(_ffi.lib().bt_tile_tree_apply_requests(self._h, self.atlas._h))

    def adjust_to_tile_atlas(self):
        _ffi.check(_ffi.lib().bt_tile_tree_adjust_to_tile_atlas(self._h, self.atlas._h))

    def read(self):
        """(entries (n, 2) u32, origins (sides, lods, 2) u32, node coordinates (n, 4) u32, requested (n,) u32)."""
        entries = np.zeros((self.nodes, 2), np.uint32)
        origins = np.zeros((self.sides, self.lod_count, 2), np.uint32)
        coords = np.zeros((self.nodes, 4), np.uint32)
        requested = np.zeros(self.nodes, np.uint32)
        _ffi.check(_ffi.lib().bt_tile_tree_read(
            self._h, entries.ctypes.data_as(C.POINTER(_ffi.TileTreeEntryC)), self.nodes,
            origins.ctypes.data_as(C.POINTER(C.c_uint32)), origins.size,
            coords.ctypes.data_as(C.POINTER(_ffi.TileCoordinateC)), requested.ctypes.data_as(C.POINTER(C.c_uint32))))
        return entries, origins, coords, requested

    def buffers(self) -> Tuple[int, int]:
        e, o = C.c_void_p(), C.c_void_p()
        _ffi.check(_ffi.lib().bt_tile_tree_buffers(self._h, C.byref(e), C.byref(o)))
        return e.value, o.value

    def sample_attachment(self, attachment_index: int, positions: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        n = len(positions)
        out = np.zeros((n, 4), np.float32)
        heights = np.zeros(n, np.float32)
        _ffi.check(_ffi.lib().bt_tile_tree_sample_attachment(
            self._h, self.atlas._h, attachment_index, positions.ctypes.data_as(C.POINTER(C.c_double)), n,
            out.ctypes.data_as(C.POINTER(C.c_float)), heights.ctypes.data_as(C.POINTER(C.c_float))))
        return out, heights

    def approximate_height(self) -> float:
        h = C.c_float()
        _ffi.check(_ffi.lib().bt_tile_tree_approximate_height(self._h, self.atlas._h, C.byref(h)))
        return h.value

    def frame_update(self, view_position: Sequence[float], prepass=None, *, unordered=False, plain=False, keep_requests=False, keep_height=False):
        """One frame of this view as ONE library call with one host synchronisation (bt_frame_update): update -> the lists
        applied to the atlas -> adjust_to_tile_atlas -> approximate_height (left on the device) -> the tiling prepass.
        Returns bt_frame_info (list lengths, the status of the apply step, the height this frame's update used)."""
        pos = (C.c_double * 3)(*view_position)
        info = _ffi.FrameInfoC()
        flags = (_ffi.FRAME_PREPASS_UNORDERED if unordered else 0) | (_ffi.FRAME_PREPASS_PLAIN if plain else 0) | \
            (_ffi.FRAME_KEEP_REQUESTS if keep_requests else 0) | (_ffi.FRAME_KEEP_HEIGHT if keep_height else 0)
        # the lists are consumed by the call (apply_requests drains them): read them through KEEP_REQUESTS when wanted
        _ffi.check(_ffi.lib().bt_frame_update(self._h, self.atlas._h, prepass._h if prepass is not None else None, pos, flags, C.byref(info)))
        return info

    def view_state(self) -> _ffi.ViewStateC:
        v = _ffi.ViewStateC()
        _ffi.check(_ffi.lib().bt_tile_tree_view_state(self._h, C.byref(v)))
        return v

    def close(self):
        if getattr(self, "_h", None):
            if device_open(getattr(getattr(self, "atlas", None), "device", None)):
                _ffi.lib().bt_tile_tree_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def sample_attachment(tile_tree: TileTree, tile_atlas: TileAtlas, attachment_index: int, sample_world_position):
    return tile_tree.sample_attachment(attachment_index, np.asarray([sample_world_position]))[0][0]


def sample_height(tile_tree: TileTree, tile_atlas: TileAtlas, sample_world_position) -> float:
    return float(tile_tree.sample_attachment(0, np.asarray([sample_world_position]))[1][0])
