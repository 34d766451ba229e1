"""bevy_terrain_amd — MI355X-native terrain-tile preprocessing and tile refinement.

The host-side mirror of the hot path of kurtkuehnert/bevy_terrain (its `prelude`, src/lib.rs:60-86,
restricted to that path): the same type names, fields, defaults and builder calls, driving
hand-written HIP kernels for gfx950 through the C ABI of include/bevy_terrain_amd.h.
Importing this package never falls back to a CPU implementation: any call that needs the
library raises if libbevy_terrain_amd.so is missing.
"""
from .terrain import (AttachmentConfig, AttachmentFormat, TerrainConfig, TerrainModel, TerrainViewConfig,
                      TileCoordinate)
from .tile_atlas import Device, TileAtlas, generate_mipmaps, tc_decode, tc_encode
from .preprocess import AssetServer, PreprocessDataset, Preprocessor, SphericalDataset
from .tiling_prepass import TilingPrepass, make_view_state
from .tile_tree import TileTree, sample_attachment, sample_height, view_state_from_config

__all__ = [
    "AttachmentConfig", "AttachmentFormat", "TerrainConfig", "TerrainModel", "TerrainViewConfig", "TileCoordinate",
    "Device", "TileAtlas", "generate_mipmaps", "tc_decode", "tc_encode",
    "AssetServer", "PreprocessDataset", "Preprocessor", "SphericalDataset",
    "TilingPrepass", "make_view_state",
    "TileTree", "sample_attachment", "sample_height", "view_state_from_config",
]

from ._ffi import BtError  # noqa: E402,F401  (status + text of a failed C call)
