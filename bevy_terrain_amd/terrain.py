"""Configuration types of the reference, same names / fields / defaults.

TerrainConfig          <- src/terrain.rs:26-56
AttachmentConfig       <- src/terrain_data/mod.rs:87-109
AttachmentFormat       <- src/terrain_data/mod.rs:37-85
TerrainViewConfig      <- src/terrain_view.rs:18-63
TerrainModel           <- src/math/terrain_model.rs:41-220 (planar and spherical; the ellipsoid projection
                          of src/math/ellipsoid.rs is CPU view maths outside the hot path)
TileCoordinate         <- src/math/coordinate.rs:155-286
"""
from __future__ import annotations

import ctypes as C
import enum
import math
from dataclasses import dataclass, field
from typing import List, Tuple

from . import _ffi

C_SQR = 0.87 * 0.87  # src/math/mod.rs:13


class AttachmentFormat(enum.Enum):
    Rgb8 = "Rgb8"
    Rgba8 = "Rgba8"
    R16 = "R16"
    Rg16 = "Rg16"

    def id(self) -> int:  # terrain_data/mod.rs:50-57
        return {"Rgb8": 5, "Rgba8": 0, "R16": 1, "Rg16": 3}[self.value]

    def pixel_size(self) -> int:  # terrain_data/mod.rs:77-84
        return {"Rgb8": 3, "Rgba8": 4, "R16": 2, "Rg16": 4}[self.value]


@dataclass
class AttachmentConfig:
    name: str = ""
    texture_size: int = 512
    border_size: int = 1
    mip_level_count: int = 1
    format: AttachmentFormat = AttachmentFormat.R16


@dataclass(frozen=True)
class TileCoordinate:
    side: int
    lod: int
    x: int
    y: int

    INVALID_VALUE = 0xFFFFFFFF

    @staticmethod
    def invalid() -> "TileCoordinate":
        v = TileCoordinate.INVALID_VALUE
        return TileCoordinate(v, v, v, v)

    @staticmethod
    def new(side, lod, x, y) -> "TileCoordinate":
        return TileCoordinate(side, lod, x, y)

    @staticmethod
    def count(lod: int) -> int:
        return 1 << lod

    def _c(self):
        return _ffi.TileCoordinateC(self.side, self.lod, self.x, self.y)

    @staticmethod
    def _from_c(c) -> "TileCoordinate":
        return TileCoordinate(c.side, c.lod, c.x, c.y)

    def parent(self) -> "TileCoordinate":
        return TileCoordinate._from_c(_ffi.lib().bt_tile_parent(self._c()))

    def children(self) -> List["TileCoordinate"]:
        out = (_ffi.TileCoordinateC * 4)()
        _ffi.lib().bt_tile_children(self._c(), out)
        return [TileCoordinate._from_c(o) for o in out]

    def neighbours(self, spherical: bool) -> List["TileCoordinate"]:
        out = (_ffi.TileCoordinateC * 8)()
        _ffi.lib().bt_tile_neighbours(self._c(), int(spherical), out)
        return [TileCoordinate._from_c(o) for o in out]

    def path(self, path: str, extension: str) -> str:
        return f"{path}/{self}.{extension}"

    def __str__(self) -> str:
        buf = C.create_string_buffer(64)
        _ffi.lib().bt_tile_name(self._c(), buf, 64)
        return buf.value.decode()


class TerrainModel:
    """Planar / spherical terrain placement (terrain_model.rs:41-115)."""

    def __init__(self, kind: str, position, scale, min_height: float, max_height: float):
        self.kind = kind
        self.translation = tuple(float(v) for v in position)
        self.scale_vec = tuple(float(v) for v in scale)
        self.min_height = float(min_height)
        self.max_height = float(max_height)

    @staticmethod
    def planar(position, side_length: float, min_height: float, max_height: float) -> "TerrainModel":
        m = TerrainModel("planar", position, (side_length,) * 3, min_height, max_height)
        m.side_length = float(side_length)
        return m

    @staticmethod
    def sphere(position, radius: float, min_height: float, max_height: float) -> "TerrainModel":
        m = TerrainModel("spherical", position, (radius,) * 3, min_height, max_height)
        m.radius = float(radius)
        return m

    @staticmethod
    def ellipsoid(position, major_axis: float, minor_axis: float, min_height: float, max_height: float) -> "TerrainModel":
        m = TerrainModel("ellipsoidal", position, (major_axis, minor_axis, major_axis), min_height, max_height)
        m.major_axis, m.minor_axis = float(major_axis), float(minor_axis)
        return m

    def is_spherical(self) -> bool:
        return self.kind != "planar"

    def side_count(self) -> int:
        return 6 if self.is_spherical() else 1

    def scale(self) -> float:  # terrain_model.rs:183-193
        if self.kind == "planar":
            return self.side_length / 2.0
        return self.radius if self.kind == "spherical" else (self.major_axis + self.minor_axis) / 2.0

    # identity rotation (the reference constructors use DQuat::IDENTITY)
    def position_local_to_world(self, local, height: float = 0.0):
        w = [self.scale_vec[i] * local[i] + self.translation[i] for i in range(3)]
        n = [self.scale_vec[i] * (local[i] if self.is_spherical() else (0.0, 1.0, 0.0)[i]) for i in range(3)]
        l = math.sqrt(sum(v * v for v in n))
        return [w[i] + height * n[i] / l for i in range(3)]

    def position_world_to_local(self, world):  # terrain_model.rs:146-174
        p = [(world[i] - self.translation[i]) / self.scale_vec[i] for i in range(3)]
        if self.kind == "planar":
            return [p[0], 0.0, p[2]]
        l = math.sqrt(sum(v * v for v in p))
        return [v / l for v in p]

    def mesh_matrices(self) -> Tuple[List[float], List[float]]:
        """mesh[0].world_from_local (3 columns + translation) and local_from_world_transpose (3x3),
        as f32-representable Python floats, what Bevy's MeshUniform carries for TerrainModel::transform()."""
        import numpy as np

        s = np.array(self.scale_vec, dtype=np.float32)
        t = np.array(self.translation, dtype=np.float32)
        wfl = [float(s[0]), 0.0, 0.0, 0.0, float(s[1]), 0.0, 0.0, 0.0, float(s[2]), float(t[0]), float(t[1]), float(t[2])]
        inv = (np.float32(1.0) / s).astype(np.float32)  # inverse of a diagonal matrix, transposed = itself
        lfwt = [float(inv[0]), 0.0, 0.0, 0.0, float(inv[1]), 0.0, 0.0, 0.0, float(inv[2])]
        return wfl, lfwt


@dataclass
class TerrainConfig:
    lod_count: int = 1
    model: TerrainModel = field(default_factory=lambda: TerrainModel.sphere((0.0, 0.0, 0.0), 1.0, 0.0, 1.0))
    atlas_size: int = 1024
    path: str = ""
    attachments: List[AttachmentConfig] = field(default_factory=list)

    def add_attachment(self, attachment_config: AttachmentConfig) -> "TerrainConfig":
        self.attachments.append(attachment_config)
        return self


@dataclass
class TerrainViewConfig:
    tree_size: int = 8
    geometry_tile_count: int = 1000000
    refinement_count: int = 30
    grid_size: int = 16
    subdivision_tolerance: float = 0.1
    precision_threshold_distance: float = 0.001
    load_distance: float = 2.5
    morph_distance: float = 16.0
    blend_distance: float = 2.0
    morph_range: float = 0.2
    blend_range: float = 0.2
    origin_lod: int = 10
