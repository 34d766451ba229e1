"""Multi-GPU preprocessing: column-strip sharding + in-place all-gather of the atlas.

New design — the reference is single-process, single-GPU (SURVEY.md §2 "Parallelism strategies: none",
§8e).  Terrain tiles are independent except parent <- children and the b-pixel aprons, so:

  1. every rank builds the SAME task queue (identical atlas indices: the integer tile-index contract),
  2. rank r preprocesses the columns [r*n/G, (r+1)*n/G) of the LODs the fused main kernel produces
     (finest three); finest-LOD aprons are evaluated from the source, so no halo exchange is needed,
  3. ONE in-place all-gather per (side, LOD) range assembles those LODs on every rank — atlas indices are
     x-major, so a rank's tiles of a LOD are one contiguous run of layers and the ranks' runs are adjacent:
     sendbuff = recvbuff + rank * count, no packing, no staging copy,
  4. every rank finishes redundantly: the cross-strip aprons of the gathered parent LODs and the few top
     LODs (< 2 % of the work).

The collective is torch.distributed (backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the CPU tests)
issued on the same stream the kernels run on.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

from . import _ffi
from .preprocess import AssetServer, PreprocessDataset, Preprocessor
from .tile_atlas import TileAtlas


class _DeviceBytes:
    """Expose a raw device allocation to torch through __cuda_array_interface__ (zero copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def shard_ranges(pre: Preprocessor) -> List[dict]:
    n = C.c_uint32()
    out = (_ffi.ShardRangeC * 64)()
    _ffi.check(_ffi.lib().bt_preprocessor_shard_ranges(pre._h, out, 64, C.byref(n)))
    return [dict(attachment_index=out[i].attachment_index, side=out[i].side, lod=out[i].lod,
                 first_layer=out[i].first_layer, layers_per_rank=out[i].layers_per_rank) for i in range(min(n.value, 64))]


def all_gather_ranges(storage, tile_bytes: int, ranges: List[dict], rank: int, world: int, dist, group=None):
    """In-place all-gather of every range over `storage` (a flat uint8 torch tensor, CPU or GPU)."""
    for r in ranges:
        count = r["layers_per_rank"] * tile_bytes
        first = r["first_layer"] * tile_bytes
        whole = storage[first:first + world * count]
        mine = whole[rank * count:(rank + 1) * count]
        dist.all_gather_into_tensor(whole, mine, group=group)


class ShardedPreprocess:
    """One sharded preprocess job: `step()` = local kernels -> all-gathers -> finishing kernels."""

    def __init__(self, pre: Preprocessor, tile_atlas: TileAtlas, asset_server: AssetServer, path: str, lod_range: range,
                 rank: int, world: int, *, attachment_index: int = 0, generic: bool = False):
        import torch
        import torch.distributed as dist

        self.pre, self.atlas, self.rank, self.world = pre, tile_atlas, rank, world
        self.dist = dist
        self.flags = (_ffi.RUN_GENERIC if generic else 0) | _ffi.RUN_KEEP_QUEUE
        pre.preprocess_tile(PreprocessDataset(attachment_index=attachment_index, path=path, lod_range=lod_range),
                            asset_server, tile_atlas)
        _ffi.check(_ffi.lib().bt_preprocessor_set_shard(pre._handle(tile_atlas), rank, world))
        ptr, tile_bytes, layers = tile_atlas.attachment_storage(attachment_index)
        self.tile_bytes = tile_bytes
        self.storage = torch.as_tensor(_DeviceBytes(ptr, tile_bytes * layers), device=f"cuda:{tile_atlas.device.index}")
        self.stream = tile_atlas.device.torch_stream
        self._ranges: Optional[List[dict]] = None
        self.gather_bytes = 0

    def _run(self, flags):
        _ffi.check(_ffi.lib().bt_preprocessor_run(self.pre._h, self.atlas._h, self.flags | flags))

    def step(self, profile: bool = False, gather: bool = True):
        """gather=False skips the collectives (timing of the kernels alone; the atlas is then incomplete)."""
        import torch

        p = _ffi.RUN_PROFILE if profile else 0
        self._run(_ffi.RUN_SHARD_LOCAL | p)
        if self._ranges is None:
            self._ranges = shard_ranges(self.pre)
            self.gather_bytes = sum(r["layers_per_rank"] * self.world * self.tile_bytes for r in self._ranges)
        if gather:
            with torch.cuda.stream(self.stream):  # same queue as the kernels: ordered without host syncs
                all_gather_ranges(self.storage, self.tile_bytes, self._ranges, self.rank, self.world, self.dist)
        self._run(_ffi.RUN_SHARD_FINISH)

    def stats(self):
        return self.pre.stats()

    def profile(self):
        return self.pre.profile()
