"""Preprocessor / PreprocessDataset / SphericalDataset (src/preprocess/preprocessor.rs) over the C ABI.

The builder API is the reference's: `Preprocessor.new().clear_attachment(i, atlas).preprocess_tile(
dataset, asset_server, atlas)`.  Where the reference then lets Bevy's schedule drain the queue over
many frames (`commands.spawn((tile_atlas, preprocessor))`), the host calls `.run(atlas)` — the whole
queue becomes a handful of kernel launches on the device's stream — and `.save(atlas, assets_root)`.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import _ffi
from .terrain import AttachmentFormat
from .tile_atlas import TileAtlas, device_open


class AssetServer:
    """Stands in for Bevy's AssetServer: `load(path)` returns the source raster registered under `path`
    (numpy array, torch CUDA tensor, or a (device_ptr, width, height, format) tuple) or decodes an image
    file below `root` with the library's PNG / TIFF decoder (bt_image_load; formats/tiff.rs forces R16Unorm the
    same way)."""

    def __init__(self, root: str = "assets"):
        self.root = root
        self._rasters: Dict[str, object] = {}

    def insert(self, path: str, raster) -> "AssetServer":
        self._rasters[path] = raster
        return self

    def load(self, path: str, fmt: Optional[AttachmentFormat] = None):
        if path in self._rasters:
            return self._rasters[path]
        full = os.path.join(self.root, path)
        if not os.path.exists(full):
            raise FileNotFoundError(f"source raster {path!r} neither registered nor found at {full}")
        if full.endswith(".npy"):
            return np.load(full)
        # PNG / TIFF: decoded by the library (bt_image_load), into the attachment's processing format; without one the file
        # decides: 16-bit grayscale -> R16, anything 8-bit -> Rgba8
        if fmt is not None:
            return decode_image(full, fmt)
        try:
            return decode_image(full, AttachmentFormat.R16)
        except _ffi.BtError as e:
            if e.status != -5:  # BT_ERR_UNSUPPORTED: not a 16-bit grayscale image
                raise
            return decode_image(full, AttachmentFormat.Rgba8)


def decode_image(path_or_bytes, fmt: AttachmentFormat) -> np.ndarray:
    """bt_image_load / bt_image_decode -> numpy array (H, W) uint16 or (H, W, 4) uint8."""
    img = _ffi.ImageC()
    if isinstance(path_or_bytes, (bytes, bytearray)):
        buf = (C.c_uint8 * len(path_or_bytes)).from_buffer_copy(bytes(path_or_bytes))
        _ffi.check(_ffi.lib().bt_image_decode(buf, len(path_or_bytes), fmt.id(), C.byref(img)))
    else:
        _ffi.check(_ffi.lib().bt_image_load(os.fsencode(path_or_bytes), fmt.id(), C.byref(img)))
    height, width = img.height, img.width
    try:
        nbytes = img.row_pitch * height
        raw = np.frombuffer((C.c_uint8 * nbytes).from_address(img.data), dtype=np.uint8).copy()
    finally:
        _ffi.lib().bt_image_free(C.byref(img))  # (zeroes the struct)
    if fmt == AttachmentFormat.R16:
        return raw.view(np.uint16).reshape(height, width)
    return raw.reshape(height, width, 4)


@dataclass
class PreprocessDataset:  # preprocessor.rs:35-55
    attachment_index: int = 0
    path: str = ""
    side: int = 0
    top_left: Tuple[float, float] = (0.0, 0.0)
    bottom_right: Tuple[float, float] = (1.0, 1.0)
    lod_range: range = range(0, 1)


@dataclass
class SphericalDataset:  # preprocessor.rs:29-33
    attachment_index: int = 0
    paths: List[str] = field(default_factory=list)
    lod_range: range = range(0, 1)


def _raster_struct(raster, fmt: AttachmentFormat, keep: list, defer_upload: bool = False) -> _ffi.RasterC:
    r = _ffi.RasterC()
    r.format = fmt.id()
    if isinstance(raster, tuple):  # (device_ptr, width, height[, row_pitch])
        r.data, r.width, r.height = raster[0], raster[1], raster[2]
        r.row_pitch = raster[3] if len(raster) > 3 else 0
        r.on_device = 1
        return r
    if hasattr(raster, "data_ptr"):  # torch tensor
        t = raster
        if t.is_cuda:
            if t.stride(-1) != 1 and not (t.dim() == 3 and t.stride(-1) == 1):
                raise ValueError("source tensor rows must be contiguous")
            keep.append(t)
            r.data, r.height, r.width = t.data_ptr(), t.shape[0], t.shape[1]
            r.row_pitch = t.stride(0) * t.element_size()
            r.on_device = 1
            return r
        raster = t.numpy()
    a = np.ascontiguousarray(raster)
    want = np.uint16 if fmt == AttachmentFormat.R16 else np.uint8
    if a.dtype != want or (fmt == AttachmentFormat.Rgba8 and (a.ndim != 3 or a.shape[2] != 4)):
        raise ValueError(f"raster dtype/shape {a.dtype}{a.shape} does not match attachment format {fmt.value}")
    keep.append(a)
    r.data, r.height, r.width = a.ctypes.data, a.shape[0], a.shape[1]
    r.row_pitch = 0
    r.on_device = _ffi.RASTER_HOST_DEFERRED if defer_upload else 0  # deferred: the rows stay ours (kept alive) until the queue has run
    return r


def _stream_stats(st) -> dict:
    d = {k: getattr(st, k) for k, _ in st._fields_}
    d["streamed"] = bool(d["streamed"])
    return d


class Preprocessor:
    def __init__(self, device=None):
        self._device = device
        self._h = None
        self._keep: list = []

    @staticmethod
    def new() -> "Preprocessor":
        return Preprocessor()

    def _handle(self, tile_atlas: TileAtlas):
        if self._h is None:
            self._device = self._device or tile_atlas.device
            h = C.c_void_p()
            _ffi.check(_ffi.lib().bt_preprocessor_create(self._device._h, C.byref(h)))
            self._h = h
        return self._h

    def clear_attachment(self, attachment_index: int, tile_atlas: TileAtlas, assets_root: Optional[str] = None) -> "Preprocessor":
        """existing_tiles.clear() and, when `assets_root` is given, reset_directory() of the attachment
        folder (preprocessor.rs:18-22, 290-296)."""
        d = tile_atlas.attachment_directory(assets_root, attachment_index).encode() if assets_root else None
        _ffi.check(_ffi.lib().bt_preprocessor_clear_attachment(self._handle(tile_atlas), tile_atlas._h, attachment_index, d))
        return self

    def preprocess_tile(self, dataset: PreprocessDataset, asset_server: AssetServer, tile_atlas: TileAtlas, *,
                        defer_upload: bool = False) -> "Preprocessor":
        """defer_upload: a host raster is not copied to the GPU by this call but when the queue runs — band by band beside the
        kernels and the downloads with run_streamed()"""
        fmt = tile_atlas.config.attachments[dataset.attachment_index].format
        raster = _raster_struct(asset_server.load(dataset.path, fmt) if isinstance(asset_server, AssetServer) else asset_server.load(dataset.path), fmt, self._keep,
                                defer_upload)
        d = _ffi.PreprocessDatasetC(dataset.attachment_index, dataset.side, (C.c_float * 2)(*dataset.top_left),
                                    (C.c_float * 2)(*dataset.bottom_right), dataset.lod_range.start, dataset.lod_range.stop)
        _ffi.check(_ffi.lib().bt_preprocessor_preprocess_tile(self._handle(tile_atlas), tile_atlas._h, C.byref(d), C.byref(raster)))
        return self

    def preprocess_spherical(self, dataset: SphericalDataset, asset_server: AssetServer, tile_atlas: TileAtlas, *,
                             defer_upload: bool = False) -> "Preprocessor":
        fmt = tile_atlas.config.attachments[dataset.attachment_index].format
        def load(path):  # (duck-typed asset servers predate the `fmt` argument: same guard as preprocess_tile)
            return asset_server.load(path, fmt) if isinstance(asset_server, AssetServer) else asset_server.load(path)

        rasters = (_ffi.RasterC * 6)(*[_raster_struct(load(p), fmt, self._keep, defer_upload) for p in dataset.paths])
        d = _ffi.SphericalDatasetC(dataset.attachment_index, dataset.lod_range.start, dataset.lod_range.stop)
        _ffi.check(_ffi.lib().bt_preprocessor_preprocess_spherical(self._handle(tile_atlas), tile_atlas._h, C.byref(d), rasters))
        return self

    def source_window(self, tile_atlas: TileAtlas, raster_index: int = 0, *, generic: bool = False):
        """((x0, y0, x1, y1), uploaded_bytes): the texels of source raster `raster_index` this preprocessor's launches read (a
        sharded fused plan: this rank's column strips + halo; else the whole raster) and the bytes of the last deferred host
        raster that actually travelled."""
        w, n = (C.c_uint32 * 4)(), C.c_uint64()
        _ffi.check(_ffi.lib().bt_preprocessor_source_window(self._handle(tile_atlas), tile_atlas._h, raster_index,
                                                            _ffi.RUN_GENERIC if generic else 0, w, C.byref(n)))
        return tuple(w), n.value

    def task_counts(self) -> Dict[str, int]:
        counts = (C.c_uint32 * 5)()
        if self._h is not None:
            _ffi.lib().bt_preprocessor_task_counts(self._h, counts)
        return dict(zip(("split", "stitch", "downsample", "save", "barrier"), list(counts)))

    def run(self, tile_atlas: TileAtlas, *, generic: bool = False, keep_queue: bool = False, sync: bool = True,
            profile: bool = False, reference_dispatch: bool = False) -> "Preprocessor":
        """reference_dispatch: leave the last texture_size % 8 rows of every tile unwritten, as the reference's dispatch of
        texture_size / 8 workgroup rows does (gpu_tile_atlas.rs:105); default: every row is processed."""
        flags = ((_ffi.RUN_GENERIC if generic else 0) | (_ffi.RUN_KEEP_QUEUE if keep_queue else 0)
                 | (_ffi.RUN_PROFILE if profile else 0) | (_ffi.RUN_REFERENCE_DISPATCH if reference_dispatch else 0))
        _ffi.check(_ffi.lib().bt_preprocessor_run(self._handle(tile_atlas), tile_atlas._h, flags))
        if sync:
            self._device.synchronize()
        if not keep_queue:
            self._keep.clear()
        return self

    def run_streamed(self, tile_atlas: TileAtlas, assets_root: str = "assets", *, generic: bool = False, keep_queue: bool = False,
                     reference_dispatch: bool = False) -> dict:
        """run() + save() as one overlapped pipeline (bt_preprocessor_run_streamed): upload of deferred host rasters, kernels,
        downloads and file writes at the same time — every fused job of the queue is banded (several attachments, the six faces of a
        cube job).  Returns bt_stream_stats as a dict: streamed, bands, banded_launches, early_tiles, uploaded_bytes, saved_bytes."""
        st = _ffi.StreamStatsC()
        flags = (_ffi.RUN_GENERIC if generic else 0) | (_ffi.RUN_KEEP_QUEUE if keep_queue else 0) | (_ffi.RUN_REFERENCE_DISPATCH if reference_dispatch else 0)
        _ffi.check(_ffi.lib().bt_preprocessor_run_streamed(self._handle(tile_atlas), tile_atlas._h, assets_root.encode(), flags, C.byref(st)))
        if not keep_queue:
            self._keep.clear()
        return _stream_stats(st)

    def run_streamed_sharded(self, tile_atlas: TileAtlas, assets_root: str = "assets", *, comm=None, local: bool = True, finish: bool = True,
                             keep_queue: bool = False) -> dict:
        """The end-to-end span of one rank of a sharded job with a distributed result (bt_preprocessor_run_streamed_sharded): this rank's
        source window uploads band by band, its finest tiles leave band by band, the two parent LODs are exchanged (`comm`: a
        shard.Communicator handle; None: the caller exchanges between a local=True, finish=False call and a local=False, finish=True
        call), the finishing kernels run and the rank's share of the lower LODs is written."""
        st = _ffi.StreamStatsC()
        flags = (_ffi.RUN_SHARD_LOCAL if local else 0) | (_ffi.RUN_SHARD_FINISH if finish else 0) | (_ffi.RUN_KEEP_QUEUE if keep_queue else 0)
        _ffi.check(_ffi.lib().bt_preprocessor_run_streamed_sharded(self._handle(tile_atlas), tile_atlas._h, comm, assets_root.encode(), flags, C.byref(st)))
        if not keep_queue and finish:
            self._keep.clear()
        return _stream_stats(st)

    def stats(self) -> Dict[str, int]:
        s = _ffi.RunStatsC()
        _ffi.check(_ffi.lib().bt_preprocessor_last_run_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in s._fields_}

    KINDS = ("split", "downsample", "stitch", "fused_main", "fused_tail", "fused_direct", "fused_todo")

    def profile(self) -> List[dict]:
        """Average device time per launch of the runs made with profile=True (hipEvents on the stream)."""
        n = C.c_uint32()
        out = (_ffi.LaunchProfileC * 256)()
        _ffi.check(_ffi.lib().bt_preprocessor_profile(self._h, out, 256, C.byref(n)))
        return [dict(kind=self.KINDS[out[i].kind], tasks=out[i].tasks, algorithmic_bytes=out[i].algorithmic_bytes,
                     avg_ms=out[i].avg_ms, samples=out[i].samples) for i in range(min(n.value, 256))]

    def save(self, tile_atlas: TileAtlas, assets_root: str = "assets") -> "Preprocessor":
        _ffi.check(_ffi.lib().bt_preprocessor_save(self._handle(tile_atlas), tile_atlas._h, assets_root.encode()))
        return self

    def close(self):
        if self._h is not None:
            # (cyclic garbage collection finalises objects in any order: a preprocessor whose context is gone already must not touch it)
            if device_open(self._device):
                _ffi.lib().bt_preprocessor_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
