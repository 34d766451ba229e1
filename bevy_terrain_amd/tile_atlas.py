"""Device context and TileAtlas (src/terrain_data/tile_atlas.rs:519-624 + gpu_tile_atlas.rs)."""
from __future__ import annotations

import ctypes as C
import os
from typing import List, Optional, Tuple

import numpy as np

from . import _ffi
from .terrain import AttachmentFormat, TerrainConfig, TileCoordinate


def device_open(device) -> bool:
    """False once the Device (bt_ctx) behind a dependent object has been closed.  Python's cyclic garbage collector finalises the objects of a
    cycle in any order (an exception traceback that holds an atlas, its preprocessor and the device is such a cycle): a dependent object whose
    context was destroyed first must not call into the library with it — its native half is then left to the process's exit."""
    return device is not None and bool(getattr(device, "_h", None))


class Device:
    """One GPU + one HIP stream (bt_ctx).  With PyTorch present the context runs on torch's current
    stream of that device, so torch.cuda.Event timing and torch.distributed collectives order with it."""

    def __init__(self, index: int = 0, stream: Optional[int] = None):
        L = _ffi.lib()
        self.torch_stream = None
        if stream is None:
            try:
                import torch

                if torch.cuda.is_available():
                    # a dedicated torch stream: kernels, torch.cuda.Event timing and torch.distributed
                    # collectives issued under `with torch.cuda.stream(device.torch_stream)` share one queue
                    torch.cuda.set_device(index)
                    self.torch_stream = torch.cuda.Stream(device=index)
                    stream = self.torch_stream.cuda_stream
            except ImportError:
                stream = None
        h = C.c_void_p()
        _ffi.check(L.bt_ctx_create(index, C.c_void_p(stream) if stream else None, C.byref(h)))
        self._h = h
        self.index = index

    def synchronize(self):
        _ffi.check(_ffi.lib().bt_ctx_synchronize(self._h))

    def set_io_threads(self, threads: int = 0) -> int:
        """bt_ctx_set_io_threads: writer / reader threads of this context's save and load paths (0 = automatic: min(16, CPUs the
        process may use)); returns the count the next save / load uses"""
        _ffi.check(_ffi.lib().bt_ctx_set_io_threads(self._h, threads))
        return int(_ffi.lib().bt_ctx_io_threads(self._h))

    def io_threads(self) -> int:
        return int(_ffi.lib().bt_ctx_io_threads(self._h))

    def trim(self) -> int:
        """bt_ctx_trim: give back the raster buffer kept for the next queue and the pinned staging buffers; bytes released"""
        freed = C.c_uint64()
        _ffi.check(_ffi.lib().bt_ctx_trim(self._h, C.byref(freed)))
        return freed.value

    def timer_begin(self):
        _ffi.check(_ffi.lib().bt_ctx_timer_begin(self._h))

    def timer_end(self) -> float:
        ms = C.c_float()
        _ffi.check(_ffi.lib().bt_ctx_timer_end(self._h, C.byref(ms)))
        return ms.value

    def malloc(self, nbytes: int) -> int:
        p = C.c_void_p()
        _ffi.check(_ffi.lib().bt_device_malloc(self._h, nbytes, C.byref(p)))
        return p.value

    def free(self, ptr: int):
        _ffi.check(_ffi.lib().bt_device_free(self._h, C.c_void_p(ptr)))

    def upload(self, array: np.ndarray) -> int:
        array = np.ascontiguousarray(array)
        ptr = self.malloc(array.nbytes)
        _ffi.check(_ffi.lib().bt_memcpy_h2d(self._h, C.c_void_p(ptr), array.ctypes.data_as(C.c_void_p), array.nbytes))
        return ptr

    def download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        _ffi.check(_ffi.lib().bt_memcpy_d2h(self._h, out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes))
        return out

    def synth_fbm_r16(self, width, height, seed, *, x0=0, y0=0, base_cell=None, octaves=6, dst: Optional[int] = None, pitch: Optional[int] = None) -> int:
        """Deterministic integer fBm heightmap in HBM; returns the device pointer (u16, tightly packed).  With dst / pitch / x0 / y0
        / base_cell it fills a WINDOW of a larger raster: texel (i, j) of the call is texel (x0 + i, y0 + j) of the pattern."""
        base_cell = base_cell or max(max(width, height) // 4, 1)
        ptr = dst if dst is not None else self.malloc(width * height * 2)
        _ffi.check(_ffi.lib().bt_synth_fbm_r16(self._h, C.c_void_p(ptr), width, height, pitch or width * 2, x0, y0, base_cell,
                                               octaves, seed & 0xFFFFFFFF))
        return ptr

    def close(self):
        if getattr(self, "_h", None):
            _ffi.lib().bt_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def texel_dtype(fmt: AttachmentFormat):
    return np.uint16 if fmt == AttachmentFormat.R16 else np.uint8


class TileAtlas:
    """TileAtlas::new(&TerrainConfig) — owns the attachment atlases in HBM and the tile index allocator."""

    def __init__(self, config: TerrainConfig, device: Optional[Device] = None):
        self.config = config
        self.device = device or Device(0)
        self.model = config.model
        self.lod_count = config.lod_count
        self.atlas_size = config.atlas_size
        self.path = config.path
        cfg = _ffi.TerrainConfigC()
        cfg.lod_count = config.lod_count
        cfg.atlas_size = config.atlas_size
        cfg.spherical = int(config.model.is_spherical())
        cfg.attachment_count = len(config.attachments)
        for i, a in enumerate(config.attachments):
            cfg.attachments[i].name = a.name.encode()[:63]
            cfg.attachments[i].texture_size = a.texture_size
            cfg.attachments[i].border_size = a.border_size
            cfg.attachments[i].mip_level_count = a.mip_level_count
            cfg.attachments[i].format = a.format.id()
        cfg.path = config.path.encode()[:255]
        h = C.c_void_p()
        _ffi.check(_ffi.lib().bt_atlas_create(self.device._h, C.byref(cfg), C.byref(h)))
        self._h = h

    @staticmethod
    def new(config: TerrainConfig, device: Optional[Device] = None) -> "TileAtlas":
        return TileAtlas(config, device)

    # --- tile_atlas.rs:553-559
    def get_tile(self, c: TileCoordinate) -> Tuple[TileCoordinate, int]:
        t = _ffi.AtlasTileC()
        _ffi.check(_ffi.lib().bt_atlas_get_tile(self._h, c._c(), C.byref(t)))
        return TileCoordinate._from_c(t.coordinate), t.atlas_index

    def get_or_allocate_tile(self, c: TileCoordinate) -> Tuple[TileCoordinate, int]:
        t = _ffi.AtlasTileC()
        _ffi.check(_ffi.lib().bt_atlas_get_or_allocate_tile(self._h, c._c(), C.byref(t)))
        return TileCoordinate._from_c(t.coordinate), t.atlas_index

    def tiles(self) -> List[Tuple[TileCoordinate, int]]:
        """existing tiles with their atlas indices, in allocation order."""
        n = _ffi.lib().bt_atlas_tiles(self._h, None, None, 0)
        coords = (_ffi.TileCoordinateC * max(n, 1))()
        idx = (C.c_uint32 * max(n, 1))()
        _ffi.lib().bt_atlas_tiles(self._h, coords, idx, n)
        return [(TileCoordinate._from_c(coords[i]), idx[i]) for i in range(n)]

    # --- the streaming half of TileAtlasState (tile_atlas.rs:418-503)
    def request_tile(self, c: TileCoordinate):
        _ffi.check(_ffi.lib().bt_atlas_request_tile(self._h, c._c()))

    def release_tile(self, c: TileCoordinate):
        _ffi.check(_ffi.lib().bt_atlas_release_tile(self._h, c._c()))

    def get_best_tile(self, c: TileCoordinate) -> Tuple[int, int]:
        e = _ffi.TileTreeEntryC()
        _ffi.check(_ffi.lib().bt_atlas_get_best_tile(self._h, c._c(), C.byref(e)))
        return e.atlas_index, e.atlas_lod

    def pending_loads(self) -> int:
        return _ffi.lib().bt_atlas_pending_loads(self._h)

    def update(self, assets_root: str = "assets", max_loads: int = 0) -> Tuple[int, int]:
        """TileAtlasState::update + AtlasAttachment::update: run the queued tile loads; (loaded, failed)."""
        loaded, failed = C.c_uint32(), C.c_uint32()
        _ffi.check(_ffi.lib().bt_atlas_update(self._h, assets_root.encode(), max_loads, C.byref(loaded), C.byref(failed)))
        return loaded.value, failed.value

    def attachment_storage(self, attachment_index: int) -> Tuple[int, int, int]:
        """(device pointer, bytes per tile, layers) of the attachment's level-0 atlas."""
        p, tb, layers = C.c_void_p(), C.c_uint64(), C.c_uint32()
        _ffi.check(_ffi.lib().bt_atlas_attachment_storage(self._h, attachment_index, C.byref(p), C.byref(tb), C.byref(layers)))
        return p.value, tb.value, layers.value

    def _tile_shape(self, attachment_index, mip=0):
        a = self.config.attachments[attachment_index]
        T = a.texture_size >> mip
        return (T, T) if a.format == AttachmentFormat.R16 else (T, T, 4)

    def download_tiles(self, attachment_index: int, first_layer: int, count: int) -> np.ndarray:
        a = self.config.attachments[attachment_index]
        out = np.empty((count,) + self._tile_shape(attachment_index), dtype=texel_dtype(a.format))
        _ffi.check(_ffi.lib().bt_atlas_download_tiles(self._h, attachment_index, first_layer, count,
                                                      out.ctypes.data_as(C.c_void_p), out.nbytes))
        return out

    def download_tile(self, attachment_index: int, atlas_index: int) -> np.ndarray:
        return self.download_tiles(attachment_index, atlas_index, 1)[0]

    def upload_tile(self, attachment_index: int, atlas_index: int, data: np.ndarray):
        data = np.ascontiguousarray(data)
        _ffi.check(_ffi.lib().bt_atlas_upload_tile(self._h, attachment_index, atlas_index, data.ctypes.data_as(C.c_void_p), data.nbytes))

    def generate_mipmaps(self, attachment_index: int, first_layer: int = 0, count: Optional[int] = None):
        count = self.atlas_size - first_layer if count is None else count
        _ffi.check(_ffi.lib().bt_atlas_generate_mipmaps(self._h, attachment_index, first_layer, count))

    def download_mip(self, attachment_index: int, mip: int, atlas_index: int) -> np.ndarray:
        p, tb = C.c_void_p(), C.c_uint64()
        _ffi.check(_ffi.lib().bt_atlas_mip_storage(self._h, attachment_index, mip, C.byref(p), C.byref(tb)))
        a = self.config.attachments[attachment_index]
        return self.device.download(p.value + tb.value * atlas_index, self._tile_shape(attachment_index, mip), texel_dtype(a.format))

    def attachment_directory(self, assets_root: str, attachment_index: int) -> str:
        """AtlasAttachment::new: "assets/{path}/data/{name}" (tile_atlas.rs:175)."""
        return os.path.join(assets_root, self.path, "data", self.config.attachments[attachment_index].name)

    def save_attachment(self, attachment_index: int, directory: str):
        _ffi.check(_ffi.lib().bt_atlas_save_attachment(self._h, attachment_index, directory.encode()))

    def save_tile_config(self, assets_root: str = "assets"):
        os.makedirs(os.path.join(assets_root, self.path), exist_ok=True)
        _ffi.check(_ffi.lib().bt_atlas_save_tile_config(self._h, os.path.join(assets_root, self.path, "config.tc").encode()))

    def load_tile_config(self, assets_root: str = "assets"):
        _ffi.check(_ffi.lib().bt_atlas_load_tile_config(self._h, os.path.join(assets_root, self.path, "config.tc").encode()))

    def load_tiles(self, attachment_index: int, assets_root: str = "assets", coords: Optional[list] = None):
        """The tile load path (start_loading + upload_tiles, tile_atlas.rs:118-149, gpu_tile_atlas.rs:309-336) for a
        batch: `.bin` files -> atlas layers, then the mip chain of those layers on the GPU.  coords=None: every tile
        of the loaded tile config."""
        directory = self.attachment_directory(assets_root, attachment_index).encode()
        if coords is None:
            _ffi.check(_ffi.lib().bt_atlas_load_tiles(self._h, attachment_index, directory, None, 0))
        else:
            arr = (_ffi.TileCoordinateC * max(len(coords), 1))(*[c._c() for c in coords])
            _ffi.check(_ffi.lib().bt_atlas_load_tiles(self._h, attachment_index, directory, arr, len(coords)))
        return self

    def sample(self, attachment_index: int, atlas_indices, atlas_uvs, atlas_lods=None) -> np.ndarray:
        """TileAtlas::sample_attachment for a batch of TileLookups (tile_atlas.rs:249-258, 569-571): bilinear sample
        of level 0 of tile `atlas_indices[i]` at `atlas_uvs[i]` (uv over the tile's centre) -> (n, 4) float32."""
        idx = np.ascontiguousarray(atlas_indices, dtype=np.uint32).ravel()
        uv = np.ascontiguousarray(atlas_uvs, dtype=np.float32).reshape(-1, 2)
        n = len(idx)
        lookups = np.zeros(n, dtype=np.dtype([("atlas_index", "<u4"), ("atlas_lod", "<u4"), ("atlas_uv", "<f4", (2,))]))
        lookups["atlas_index"] = idx
        lookups["atlas_lod"] = 0 if atlas_lods is None else np.asarray(atlas_lods, dtype=np.uint32)
        lookups["atlas_uv"] = uv
        out = np.empty((n, 4), dtype=np.float32)
        _ffi.check(_ffi.lib().bt_atlas_sample(self._h, attachment_index, lookups.ctypes.data_as(C.c_void_p), n,
                                              out.ctypes.data_as(C.c_void_p)))
        return out

    def close(self):
        if getattr(self, "_h", None):
            if device_open(getattr(self, "device", None)):
                _ffi.lib().bt_atlas_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def generate_mipmaps(device: Device, fmt: AttachmentFormat, level0: np.ndarray, mip_level_count: int) -> np.ndarray:
    """AttachmentData::generate_mipmaps (terrain_data/mod.rs:143-219) on the GPU: all levels concatenated."""
    T = level0.shape[0]
    ch = 1 if fmt == AttachmentFormat.R16 else 4
    total = sum((T >> k) ** 2 for k in range(mip_level_count)) * ch
    level0 = np.ascontiguousarray(level0)
    out = np.empty(total, dtype=texel_dtype(fmt))
    _ffi.check(_ffi.lib().bt_generate_mipmaps(device._h, fmt.id(), T, mip_level_count, level0.ctypes.data_as(C.c_void_p),
                                              out.ctypes.data_as(C.c_void_p), out.nbytes))
    return out


def tc_encode(coords: List[TileCoordinate]) -> bytes:
    arr = (_ffi.TileCoordinateC * max(len(coords), 1))(*[c._c() for c in coords])
    n = _ffi.lib().bt_tc_encode(arr, len(coords), None, 0)
    buf = (C.c_uint8 * n)()
    _ffi.lib().bt_tc_encode(arr, len(coords), buf, n)
    return bytes(buf)


def tc_decode(data: bytes) -> List[TileCoordinate]:
    buf = (C.c_uint8 * max(len(data), 1)).from_buffer_copy(data or b"\0")
    n = _ffi.lib().bt_tc_decode(buf, len(data), None, 0)
    if n < 0:
        raise ValueError("malformed tile config")
    out = (_ffi.TileCoordinateC * max(n, 1))()
    _ffi.lib().bt_tc_decode(buf, len(data), out, n)
    return [TileCoordinate._from_c(out[i]) for i in range(n)]
