"""Per-frame tile refinement (UDLOD tiling prepass) behind the reference's view types.

TerrainViewData / TilingPrepassNode  <- src/render/terrain_view_bind_group.rs:118-247,
                                        src/render/tiling_prepass.rs:204-272
view uniform derivation              <- TileTree::new (src/terrain_data/tile_tree.rs:135-173),
                                        TerrainViewConfigUniform::from_tile_tree (terrain_view_bind_group.rs:98-116)
view coordinate per cube side        <- Coordinate::from_world_position / project_to_side
                                        (src/math/coordinate.rs:69-151), TerrainModelApproximation::compute
                                        (src/math/terrain_model.rs:262-290; only view_xy / view_uv are read by
                                        the prepass because HIGH_PRECISION is never defined for it)
The f64 host maths here is glue that feeds the kernel; the refinement itself runs in bt_refine.hip.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _ffi
from .terrain import C_SQR, TerrainModel, TerrainViewConfig, TileCoordinate
from .tile_atlas import Device, device_open

# SideInfo tables of coordinate.rs:27-42 (0 = Fixed0, 1 = Fixed1, 's' / 't')
_EVEN = [("s", "t"), (0, "t"), (0, "s"), ("t", "s"), ("t", 0), ("s", 0)]
_ODD = [("s", "t"), ("s", 1), ("t", 1), ("t", "s"), (1, "s"), (1, "t")]


def coordinate_from_world_position(world_position: Sequence[float], model: TerrainModel) -> Tuple[int, Tuple[float, float]]:
    """Coordinate::from_world_position (coordinate.rs:69-113), f64."""
    p = model.position_world_to_local(world_position)
    if not model.is_spherical():
        return 0, (min(max(p[0] + 0.5, 0.0), 1.0), min(max(p[2] + 0.5, 0.0), 1.0))
    nx, ny, nz = p
    ax, ay, az = abs(nx), abs(ny), abs(nz)
    if ax > ay and ax > az:
        side, uv = (0, (-nz / nx, ny / nx)) if nx < 0.0 else (3, (-ny / nx, nz / nx))
    elif az > ay:
        side, uv = (1, (nx / nz, -ny / nz)) if nz > 0.0 else (4, (ny / nz, -nx / nz))
    else:
        side, uv = (2, (nx / ny, nz / ny)) if ny > 0.0 else (5, (-nz / ny, -nx / ny))
    w = [u * math.sqrt((1.0 + C_SQR) / (1.0 + C_SQR * u * u)) for u in uv]
    return side, (0.5 * w[0] + 0.5, 0.5 * w[1] + 0.5)


def project_to_side(side: int, uv: Tuple[float, float], other: int, model: TerrainModel) -> Tuple[float, float]:
    """Coordinate::project_to_side (coordinate.rs:137-151)."""
    if not model.is_spherical():
        return uv
    info = (_EVEN if side % 2 == 0 else _ODD)[(6 + other - side) % 6]
    pick = lambda i: uv[0] if i == "s" else uv[1] if i == "t" else float(i)
    return pick(info[0]), pick(info[1])


def make_view_state(model: TerrainModel, view_config: TerrainViewConfig, view_world_position: Sequence[float], *,
                    approximate_height: Optional[float] = None) -> _ffi.ViewStateC:
    """Everything `refine_tiles` reads for one view and frame: derived by the library (bt_view_state_from_config,
    f64 on the host like TileTree::new / TerrainViewConfigUniform::from_tile_tree /
    TerrainModelApproximation::compute).  approximate_height defaults to TileTree::new's (min + max) / 2."""
    from .tile_tree import view_state_from_config

    height = (model.min_height + model.max_height) / 2.0 if approximate_height is None else approximate_height
    return view_state_from_config(model, view_config, view_world_position, float(np.float32(height)))


class TilingPrepass:
    """TerrainViewData buffers + TilingPrepassNode::run as one persistent launch."""

    def __init__(self, device: Device, geometry_tile_count: int = 1000000):
        self.device = device
        self.capacity = geometry_tile_count
        h = C.c_void_p()
        _ffi.check(_ffi.lib().bt_tiling_prepass_create(device._h, geometry_tile_count, C.byref(h)))
        self._h = h

    def run(self, view: _ffi.ViewStateC, *, plain: bool = False, unordered: bool = False):
        """default: two launches — the divide tests of every tile near the view at every LOD up front (chip-wide), then the
        ordered schedule over those bits.  plain=True: one launch that evaluates each test inside its pass.  Both: the list in
        the order of the reference run with invocations taken in id order.  unordered=True: the same SET (the reference's
        contract: its order is the arrival order of an atomic) from two chip-wide launches without the chain of passes."""
        if unordered:
            _ffi.check(_ffi.lib().bt_tiling_prepass_run_unordered(self._h, C.byref(view)))
        elif plain:
            _ffi.check(_ffi.lib().bt_tiling_prepass_run_plain(self._h, C.byref(view)))
        else:
            _ffi.check(_ffi.lib().bt_tiling_prepass_run(self._h, C.byref(view)))

    def set_window(self, radius: int):
        """window radius of the unordered form (1..28, 0 = default): a tuning / test knob, results do not depend on it"""
        _ffi.check(_ffi.lib().bt_tiling_prepass_set_window(self._h, radius))

    def read(self) -> Tuple[np.ndarray, Tuple[int, int, int, int]]:
        """(final tiles as an (n, 4) uint32 array [side, lod, x, y] in append order, indirect draw args)."""
        n = C.c_uint32()
        ind = _ffi.IndirectC()
        _ffi.check(_ffi.lib().bt_tiling_prepass_read(self._h, None, 0, C.byref(n), C.byref(ind)))
        out = np.zeros((max(n.value, 1), 4), dtype=np.uint32)
        _ffi.check(_ffi.lib().bt_tiling_prepass_read(self._h, out.ctypes.data_as(C.POINTER(_ffi.TileCoordinateC)), n.value,
                                                     C.byref(n), C.byref(ind)))
        return out[: n.value], (ind.vertex_count, ind.instance_count, ind.base_vertex, ind.base_instance)

    def buffers(self) -> Tuple[int, int]:
        f, i = C.c_void_p(), C.c_void_p()
        _ffi.check(_ffi.lib().bt_tiling_prepass_buffers(self._h, C.byref(f), C.byref(i)))
        return f.value, i.value

    def close(self):
        if getattr(self, "_h", None):
            if device_open(getattr(self, "device", None)):
                _ffi.lib().bt_tiling_prepass_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
