"""ctypes binding of libbevy_terrain_amd.so (the C ABI in include/bevy_terrain_amd.h).

The product path has no CPU fallback: if the HIP library is missing or fails to load this module
raises, loudly.  PyTorch (when installed) is imported first so that the library binds to the same
HIP runtime instance torch uses — device pointers and streams are then interchangeable.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libbevy_terrain_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_HERE), "include", "bevy_terrain_amd.h")


def header_abi_version() -> int:
    """BT_ABI_VERSION as include/bevy_terrain_amd.h declares it: the one place the number is written."""
    import re
    try:
        with open(HEADER_PATH) as f:
            text = f.read()
    except OSError as e:  # a deployment without the sibling include/ directory: a load failure like any other
        raise ImportError(f"bevy_terrain_amd: {HEADER_PATH} (the C ABI's header, which names the ABI version this binding checks the "
                          f"library against) cannot be read: {e}") from e
    m = re.search(r"^#define\s+BT_ABI_VERSION\s+(\d+)u?\s*$", text, flags=re.M)
    if not m:
        raise ImportError(f"{HEADER_PATH}: BT_ABI_VERSION not found")
    return int(m.group(1))


def header_symbols() -> set:
    """Every bt_* function include/bevy_terrain_amd.h declares."""
    import re
    with open(HEADER_PATH) as f:
        text = re.sub(r"/\*.*?\*/", "", f.read(), flags=re.S)
    return set(re.findall(r"\b(bt_[a-z0-9_]+)\s*\(", text))

INVALID_ATLAS_INDEX = 0xFFFFFFFF
MAX_ATTACHMENTS = 8

BT_OK = 0
RUN_AUTO, RUN_GENERIC, RUN_KEEP_QUEUE, RUN_PROFILE, RUN_SHARD_LOCAL, RUN_SHARD_FINISH, RUN_SHARD_DISTRIBUTED, RUN_SHARD_EXCHANGE, RUN_SHARD_OVERLAP, RUN_REFERENCE_DISPATCH = 0, 1, 2, 4, 8, 16, 32, 64, 128, 256


class BtError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"bevy_terrain_amd: status {status}: {message}")
        self.status = status


class TileCoordinateC(C.Structure):
    _fields_ = [("side", C.c_uint32), ("lod", C.c_uint32), ("x", C.c_uint32), ("y", C.c_uint32)]


class AtlasTileC(C.Structure):
    _fields_ = [("coordinate", TileCoordinateC), ("atlas_index", C.c_uint32), ("_padding", C.c_uint32 * 3)]


class AttachmentConfigC(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("texture_size", C.c_uint32), ("border_size", C.c_uint32),
                ("mip_level_count", C.c_uint32), ("format", C.c_uint32)]


class TerrainConfigC(C.Structure):
    _fields_ = [("lod_count", C.c_uint32), ("atlas_size", C.c_uint32), ("spherical", C.c_uint32),
                ("attachment_count", C.c_uint32), ("attachments", AttachmentConfigC * MAX_ATTACHMENTS),
                ("path", C.c_char * 256)]


class RasterC(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("row_pitch", C.c_uint64),
                ("format", C.c_uint32), ("on_device", C.c_uint32)]


class ImageC(C.Structure):
    _fields_ = [("data", C.c_void_p), ("width", C.c_uint32), ("height", C.c_uint32), ("format", C.c_uint32),
                ("row_pitch", C.c_uint64)]


class PreprocessDatasetC(C.Structure):
    _fields_ = [("attachment_index", C.c_uint32), ("side", C.c_uint32), ("top_left", C.c_float * 2),
                ("bottom_right", C.c_float * 2), ("lod_begin", C.c_uint32), ("lod_end", C.c_uint32)]


class SphericalDatasetC(C.Structure):
    _fields_ = [("attachment_index", C.c_uint32), ("lod_begin", C.c_uint32), ("lod_end", C.c_uint32)]


class RunStatsC(C.Structure):
    _fields_ = [("kernel_launches", C.c_uint32), ("tiles", C.c_uint32), ("algorithmic_bytes", C.c_uint64),
                ("fused_jobs", C.c_uint32), ("generic_jobs", C.c_uint32), ("prev_zero_launches", C.c_uint32), ("reserved", C.c_uint32)]


class ShardRangeC(C.Structure):
    _fields_ = [("attachment_index", C.c_uint32), ("side", C.c_uint32), ("lod", C.c_uint32), ("first_layer", C.c_uint32),
                ("layers_per_rank", C.c_uint32)]


class ShardPieceC(C.Structure):
    _fields_ = [("attachment_index", C.c_uint32), ("side", C.c_uint32), ("lod", C.c_uint32), ("first_layer", C.c_uint32),
                ("layers", C.c_uint32), ("owner_rank", C.c_uint32)]


class LaunchProfileC(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("tasks", C.c_uint32), ("algorithmic_bytes", C.c_uint64), ("avg_ms", C.c_float),
                ("samples", C.c_uint32)]


class SideParameterC(C.Structure):
    _fields_ = [("view_xy", C.c_int32 * 2), ("view_uv", C.c_float * 2)]


class ViewStateC(C.Structure):
    _fields_ = [("spherical", C.c_uint32), ("geometry_tile_count", C.c_uint32), ("refinement_count", C.c_uint32),
                ("vertices_per_tile", C.c_uint32), ("subdivision_distance", C.c_float), ("origin_lod", C.c_uint32),
                ("approximate_height", C.c_float), ("sides", SideParameterC * 6), ("world_position", C.c_float * 3),
                ("world_from_local", C.c_float * 12), ("local_from_world_transpose", C.c_float * 9)]


class IndirectC(C.Structure):
    _fields_ = [("vertex_count", C.c_uint32), ("instance_count", C.c_uint32), ("base_vertex", C.c_uint32),
                ("base_instance", C.c_uint32)]


class FrameInfoC(C.Structure):
    _fields_ = [("released_count", C.c_uint32), ("requested_count", C.c_uint32), ("apply_status", C.c_int32), ("approximate_height", C.c_float)]


FRAME_PREPASS_UNORDERED, FRAME_PREPASS_PLAIN, FRAME_KEEP_REQUESTS, FRAME_KEEP_HEIGHT = 1, 2, 4, 8


class TileTreeEntryC(C.Structure):
    _fields_ = [("atlas_index", C.c_uint32), ("atlas_lod", C.c_uint32)]


class TerrainModelC(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("_padding", C.c_uint32), ("position", C.c_double * 3), ("a", C.c_double),
                ("b", C.c_double), ("min_height", C.c_float), ("max_height", C.c_float)]


class TerrainViewConfigC(C.Structure):
    _fields_ = [("tree_size", C.c_uint32), ("geometry_tile_count", C.c_uint32), ("refinement_count", C.c_uint32),
                ("grid_size", C.c_uint32), ("subdivision_tolerance", C.c_double),
                ("precision_threshold_distance", C.c_double), ("load_distance", C.c_double),
                ("morph_distance", C.c_double), ("blend_distance", C.c_double), ("morph_range", C.c_float),
                ("blend_range", C.c_float), ("origin_lod", C.c_uint32), ("_padding", C.c_uint32)]


_vp, _u32, _u64, _i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int32
_P = C.POINTER

# name -> (restype, argtypes); every function include/bevy_terrain_amd.h declares
PROTOTYPES = {
    "bt_abi_version": (_u32, []),
    "bt_last_error": (C.c_char_p, []),
    "bt_ctx_create": (_i32, [_i32, _vp, _P(_vp)]),
    "bt_ctx_destroy": (None, [_vp]),
    "bt_ctx_set_stream": (_i32, [_vp, _vp]),
    "bt_ctx_stream": (_vp, [_vp]),
    "bt_ctx_synchronize": (_i32, [_vp]),
    "bt_ctx_trim": (_i32, [_vp, C.POINTER(C.c_uint64)]),
    "bt_ctx_set_io_threads": (_i32, [_vp, C.c_uint32]),
    "bt_ctx_io_threads": (C.c_uint32, [_vp]),
    "bt_ctx_timer_begin": (_i32, [_vp]),
    "bt_ctx_timer_end": (_i32, [_vp, _P(C.c_float)]),
    "bt_device_malloc": (_i32, [_vp, C.c_size_t, _P(_vp)]),
    "bt_device_free": (_i32, [_vp, _vp]),
    "bt_memcpy_h2d": (_i32, [_vp, _vp, _vp, C.c_size_t]),
    "bt_memcpy_d2h": (_i32, [_vp, _vp, _vp, C.c_size_t]),
    "bt_tile_children": (None, [TileCoordinateC, _P(TileCoordinateC)]),
    "bt_tile_neighbours": (None, [TileCoordinateC, _u32, _P(TileCoordinateC)]),
    "bt_tile_parent": (TileCoordinateC, [TileCoordinateC]),
    "bt_tile_name": (_i32, [TileCoordinateC, C.c_char_p, C.c_size_t]),
    "bt_atlas_create": (_i32, [_vp, _P(TerrainConfigC), _P(_vp)]),
    "bt_atlas_destroy": (None, [_vp]),
    "bt_atlas_get_tile": (_i32, [_vp, TileCoordinateC, _P(AtlasTileC)]),
    "bt_atlas_get_or_allocate_tile": (_i32, [_vp, TileCoordinateC, _P(AtlasTileC)]),
    "bt_atlas_request_tile": (_i32, [_vp, TileCoordinateC]),
    "bt_atlas_release_tile": (_i32, [_vp, TileCoordinateC]),
    "bt_atlas_get_best_tile": (_i32, [_vp, TileCoordinateC, _P(TileTreeEntryC)]),
    "bt_atlas_update": (_i32, [_vp, C.c_char_p, _u32, _P(_u32), _P(_u32)]),
    "bt_atlas_pending_loads": (_u32, [_vp]),
    "bt_atlas_tiles": (_u32, [_vp, _P(TileCoordinateC), _P(_u32), _u32]),
    "bt_atlas_attachment_storage": (_i32, [_vp, _u32, _P(_vp), _P(_u64), _P(_u32)]),
    "bt_atlas_download_tiles": (_i32, [_vp, _u32, _u32, _u32, _vp, _u64]),
    "bt_atlas_upload_tile": (_i32, [_vp, _u32, _u32, _vp, _u64]),
    "bt_atlas_save_attachment": (_i32, [_vp, _u32, C.c_char_p]),
    "bt_atlas_save_tile_config": (_i32, [_vp, C.c_char_p]),
    "bt_atlas_load_tile_config": (_i32, [_vp, C.c_char_p]),
    "bt_atlas_load_tiles": (_i32, [_vp, _u32, C.c_char_p, _vp, _u32]),
    "bt_atlas_sample": (_i32, [_vp, _u32, _vp, _u32, _vp]),
    "bt_tc_encode": (_u64, [_P(TileCoordinateC), _u32, _vp, _u64]),
    "bt_tc_decode": (C.c_int64, [_vp, _u64, _P(TileCoordinateC), _u32]),
    "bt_generate_mipmaps": (_i32, [_vp, _u32, _u32, _u32, _vp, _vp, _u64]),
    "bt_atlas_generate_mipmaps": (_i32, [_vp, _u32, _u32, _u32]),
    "bt_atlas_mip_storage": (_i32, [_vp, _u32, _u32, _P(_vp), _P(_u64)]),
    "bt_image_load": (_i32, [C.c_char_p, _u32, _P(ImageC)]),
    "bt_image_decode": (_i32, [_vp, C.c_size_t, _u32, _P(ImageC)]),
    "bt_image_free": (None, [_P(ImageC)]),
    "bt_preprocessor_create": (_i32, [_vp, _P(_vp)]),
    "bt_preprocessor_destroy": (None, [_vp]),
    "bt_preprocessor_clear_attachment": (_i32, [_vp, _vp, _u32, C.c_char_p]),
    "bt_preprocessor_preprocess_tile": (_i32, [_vp, _vp, _P(PreprocessDatasetC), _P(RasterC)]),
    "bt_preprocessor_preprocess_spherical": (_i32, [_vp, _vp, _P(SphericalDatasetC), _P(RasterC)]),
    "bt_preprocessor_task_counts": (_u32, [_vp, _P(_u32)]),
    "bt_preprocessor_run": (_i32, [_vp, _vp, _u32]),
    "bt_preprocessor_save": (_i32, [_vp, _vp, C.c_char_p]),
    "bt_preprocessor_last_run_stats": (_i32, [_vp, _P(RunStatsC)]),
    "bt_preprocessor_set_shard": (_i32, [_vp, _u32, _u32]),
    "bt_preprocessor_shard_ranges": (_i32, [_vp, _P(ShardRangeC), _u32, _P(_u32)]),
    "bt_preprocessor_shard_pieces": (_i32, [_vp, _P(ShardPieceC), _u32, _P(_u32)]),
    "bt_comm_unique_id": (_i32, [_P(C.c_uint8)]),
    "bt_comm_create": (_i32, [_vp, _u32, _u32, _P(C.c_uint8), _P(_vp)]),
    "bt_comm_adopt": (_i32, [_vp, _vp, _u32, _u32, _P(_vp)]),
    "bt_comm_destroy": (None, [_vp]),
    "bt_comm_check": (_i32, [_vp]),
    "bt_comm_preflight": (_i32, [_vp, C.c_uint64, C.POINTER(C.c_float)]),
    "bt_preprocessor_run_sharded": (_i32, [_vp, _vp, _vp, _u32]),
    "bt_preprocessor_finish_sharded": (_i32, [_vp, _vp, _vp, _u32]),
    "bt_preprocessor_source_window": (_i32, [_vp, _vp, _u32, _u32, _P(C.c_uint32), _P(C.c_uint64)]),
    "bt_preprocessor_profile": (_i32, [_vp, _P(LaunchProfileC), _u32, _P(_u32)]),
    "bt_tiling_prepass_create": (_i32, [_vp, _u32, _P(_vp)]),
    "bt_tiling_prepass_destroy": (None, [_vp]),
    "bt_tiling_prepass_run": (_i32, [_vp, _P(ViewStateC)]),
    "bt_tiling_prepass_run_plain": (_i32, [_vp, _P(ViewStateC)]),
    "bt_tiling_prepass_run_unordered": (_i32, [_vp, _P(ViewStateC)]),
    "bt_tiling_prepass_set_window": (_i32, [_vp, _u32]),
    "bt_tiling_prepass_buffers": (_i32, [_vp, _P(_vp), _P(_vp)]),
    "bt_tiling_prepass_read": (_i32, [_vp, _P(TileCoordinateC), _u32, _P(_u32), _P(IndirectC)]),
    "bt_terrain_view_config_default": (None, [_P(TerrainViewConfigC)]),
    "bt_view_state_from_config": (_i32, [_P(TerrainModelC), _P(TerrainViewConfigC), _P(C.c_double), C.c_float, _P(ViewStateC)]),
    "bt_tile_tree_create": (_i32, [_vp, _P(TerrainModelC), _u32, _P(TerrainViewConfigC), _P(_vp)]),
    "bt_tile_tree_destroy": (None, [_vp]),
    "bt_tile_tree_update": (_i32, [_vp, _P(C.c_double)]),
    "bt_tile_tree_requests": (_i32, [_vp, _P(_P(TileCoordinateC)), _P(_u32), _P(_P(TileCoordinateC)), _P(_u32)]),
    "bt_tile_tree_apply_requests": (_i32, [_vp, _vp]),
    "bt_tile_tree_adjust_to_tile_atlas": (_i32, [_vp, _vp]),
    "bt_tile_tree_buffers": (_i32, [_vp, _P(_vp), _P(_vp)]),
    "bt_tile_tree_read": (_i32, [_vp, _P(TileTreeEntryC), _u32, _P(_u32), _u32, _P(TileCoordinateC), _P(_u32)]),
    "bt_tile_tree_sample_attachment": (_i32, [_vp, _vp, _u32, _P(C.c_double), _u32, _P(C.c_float), _P(C.c_float)]),
    "bt_tile_tree_approximate_height": (_i32, [_vp, _vp, _P(C.c_float)]),
    "bt_tile_tree_view_state": (_i32, [_vp, _P(ViewStateC)]),
    "bt_frame_update": (_i32, [_vp, _vp, _vp, _P(C.c_double), C.c_uint32, _P(FrameInfoC)]),
    "bt_selftest": (_i32, [_vp, _P(_u32)]),
    "bt_synth_fbm_r16": (_i32, [_vp, _vp, _u32, _u32, _u64, _u32, _u32, _u32, _u32, _u32]),
    "bt_preprocessor_run_streamed": (_i32, [_vp, _vp, C.c_char_p, _u32, _vp]),
    "bt_preprocessor_run_streamed_sharded": (_i32, [_vp, _vp, _vp, C.c_char_p, _u32, _vp]),
}


class StreamStatsC(C.Structure):
    _fields_ = [("streamed", C.c_uint32), ("bands", C.c_uint32), ("banded_launches", C.c_uint32), ("early_tiles", C.c_uint32),
                ("uploaded_bytes", C.c_uint64), ("saved_bytes", C.c_uint64)]


RASTER_HOST_DEFERRED = 2

_lib = None


def lib():
    """Load the HIP library (once).  Raises if it is missing: there is no fallback path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C bevy_terrain_amd/csrc`). bevy_terrain_amd has no CPU fallback.")
    try:  # bind to torch's HIP runtime when torch is present (see module docstring)
        import torch  # noqa: F401
    except Exception:
        pass
    L = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (restype, argtypes) in PROTOTYPES.items():
        fn = getattr(L, name)  # AttributeError = ABI mismatch, also loud
        fn.restype = restype
        fn.argtypes = argtypes
    if L.bt_abi_version() != header_abi_version():
        raise ImportError(f"libbevy_terrain_amd.so is ABI {L.bt_abi_version()}, include/bevy_terrain_amd.h is "
                          f"{header_abi_version()}: rebuild (make -C bevy_terrain_amd/csrc)")
    _lib = L
    return L


def check(status):
    if status != BT_OK:
        raise BtError(status, lib().bt_last_error().decode(errors="replace"))
