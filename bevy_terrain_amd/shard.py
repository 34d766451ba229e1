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

Ownership goes by units (one column strip of one cube side at the granularity of the coarsest LOD the main kernel
produces): planar jobs shard into `world` equal column blocks (one in-place all-gather per LOD), the 6-face cube job
into 24 units, 24 / world per rank (one in-place broadcast per contiguous piece).  Either way the exchange of a step
is ONE grouped collective issued by the library on the kernels' stream through its own RCCL communicator
(bt_preprocessor_run_sharded, `collective="library"`); `collective="torch"` issues the same pieces through
torch.distributed instead (backend "nccl" = RCCL, or "gloo" in the CPU / single-GPU tests).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional

from . import _ffi
from .preprocess import AssetServer, PreprocessDataset, Preprocessor
from .tile_atlas import TileAtlas, device_open


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


def shard_pieces(pre: Preprocessor) -> List[dict]:
    n = C.c_uint32()
    _ffi.check(_ffi.lib().bt_preprocessor_shard_pieces(pre._h, None, 0, C.byref(n)))
    out = (_ffi.ShardPieceC * max(n.value, 1))()
    _ffi.check(_ffi.lib().bt_preprocessor_shard_pieces(pre._h, out, n.value, C.byref(n)))
    return [dict(attachment_index=out[i].attachment_index, side=out[i].side, lod=out[i].lod, first_layer=out[i].first_layer,
                 layers=out[i].layers, owner_rank=out[i].owner_rank) for i in range(n.value)]


def broadcast_pieces(storage, tile_bytes: int, pieces: List[dict], dist, group=None):
    """In-place broadcast of every piece from its owner over `storage` (a flat uint8 torch tensor, CPU or GPU)."""
    for p in pieces:
        first = p["first_layer"] * tile_bytes
        dist.broadcast(storage[first:first + p["layers"] * tile_bytes], src=p["owner_rank"], group=group)


def all_gather_ranges(storage, tile_bytes: int, ranges: List[dict], rank: int, world: int, dist, group=None):
    """In-place all-gather of every range over `storage` (a flat uint8 torch tensor, CPU or GPU)."""
    for r in ranges:
        count = r["layers_per_rank"] * tile_bytes
        first = r["first_layer"] * tile_bytes
        whole = storage[first:first + world * count]
        mine = whole[rank * count:(rank + 1) * count]
        dist.all_gather_into_tensor(whole, mine, group=group)


class ShardedPreprocess:
    """One sharded preprocess job: `step()` = local kernels -> the exchange -> finishing kernels.
    `path`: a source path (planar, preprocess_tile) or a list of six (cube, preprocess_spherical)."""

    def __init__(self, pre: Preprocessor, tile_atlas: TileAtlas, asset_server: AssetServer, path, lod_range: range,
                 rank: int, world: int, *, attachment_index: int = 0, generic: bool = False, collective: str = "torch",
                 result: str = "replicated", comm=None, dist=None, defer_upload: bool = False):
        """defer_upload: host rasters travel when the first step runs — and then only the window this rank's launches read (its
        column strips + halo: Preprocessor.source_window).
        result="replicated": every rank ends with the full atlas.  result="distributed" (planar jobs): the finest LOD is
        not exchanged — its tiles stay on the rank that computed them, only the two parent LODs travel (a quarter of the
        bytes); every rank still holds every lower LOD, and Preprocessor.save writes each rank's share."""
        import torch

        if dist is None:  # (tests emulate several ranks in one process with a stand-in)
            import torch.distributed as dist

        from .preprocess import SphericalDataset

        self.pre, self.atlas, self.rank, self.world = pre, tile_atlas, rank, world
        self.dist = dist
        self.collective = collective
        assert result in ("replicated", "distributed")
        self.result = result
        self.flags = (_ffi.RUN_GENERIC if generic else 0) | _ffi.RUN_KEEP_QUEUE | (_ffi.RUN_SHARD_DISTRIBUTED if result == "distributed" else 0)
        if isinstance(path, (list, tuple)):
            pre.preprocess_spherical(SphericalDataset(attachment_index=attachment_index, paths=list(path), lod_range=lod_range),
                                     asset_server, tile_atlas, defer_upload=defer_upload)
        else:
            pre.preprocess_tile(PreprocessDataset(attachment_index=attachment_index, path=path, lod_range=lod_range),
                                asset_server, tile_atlas, defer_upload=defer_upload)
        _ffi.check(_ffi.lib().bt_preprocessor_set_shard(pre._handle(tile_atlas), rank, world))
        a = tile_atlas.config.attachments[attachment_index]
        self.tile_bytes = a.texture_size * a.texture_size * a.format.pixel_size()
        self._attachment_index = attachment_index
        self._storage = None  # the atlas as a flat torch tensor: taken when a torch collective first needs it (handing the pointer out counts as a write to every layer)
        self.stream = tile_atlas.device.torch_stream
        self._ranges: Optional[List[dict]] = None
        self._pieces: Optional[List[dict]] = None
        self.held: Optional[List[dict]] = None  # distributed result: the finest-LOD pieces this rank keeps
        self.gather_bytes = 0
        self._comm = None
        self._owns_comm = False
        self._side_stream = None   # torch path of the overlapped step: the collectives' own queue
        self._local_done = self._exchange_done = None
        self.pending = False       # begin_step() issued, finish_step() has to follow
        if collective == "library" and comm is not None:
            self._comm = comm      # shared with another job of this rank (the overlapped pair)
        elif collective == "library":
            # the library's own RCCL communicator: rank 0 draws the unique id, torch.distributed only ships it
            uid = (C.c_uint8 * 128)()
            if rank == 0:
                _ffi.check(_ffi.lib().bt_comm_unique_id(uid))
            box = [bytes(uid)]
            if world > 1:
                dist.broadcast_object_list(box, src=0)
            uid = (C.c_uint8 * 128).from_buffer_copy(box[0])
            h = C.c_void_p()
            _ffi.check(_ffi.lib().bt_comm_create(tile_atlas.device._h, world, rank, uid, C.byref(h)))
            self._comm = h
            self._owns_comm = True

    @property
    def storage(self):
        import torch

        if self._storage is None:
            ptr, tile_bytes, layers = self.atlas.attachment_storage(self._attachment_index)
            assert tile_bytes == self.tile_bytes
            self._storage = torch.as_tensor(_DeviceBytes(ptr, tile_bytes * layers), device=f"cuda:{self.atlas.device.index}")
        return self._storage

    def run_streamed(self, assets_root: str) -> dict:
        """This rank's end-to-end span of the job (bt_preprocessor_run_streamed_sharded; result="distributed", rasters queued with
        defer_upload=True): its source window uploads band by band, its finest tiles leave band by band, the two parent LODs are
        exchanged (the library's communicator inside the one call; torch.distributed between the two halves otherwise), the finishing
        kernels run and its share of the lower LODs is written.  Returns the stream stats of the call(s), summed."""
        import torch

        assert self.result == "distributed", "a streamed sharded run writes each rank's share: result='distributed'"
        if self._comm is not None:
            st = self.pre.run_streamed_sharded(self.atlas, assets_root, comm=self._comm, keep_queue=True)
            self._layout()
            return st
        st = self.pre.run_streamed_sharded(self.atlas, assets_root, local=True, finish=False)
        self._layout()
        if self.world > 1:
            with torch.cuda.stream(self.stream):  # same queue as the kernels: ordered without host syncs
                if self._ranges:
                    all_gather_ranges(self.storage, self.tile_bytes, self._ranges, self.rank, self.world, self.dist)
                else:
                    broadcast_pieces(self.storage, self.tile_bytes, self._pieces, self.dist)
        st2 = self.pre.run_streamed_sharded(self.atlas, assets_root, local=False, finish=True, keep_queue=True)
        for k in ("uploaded_bytes", "saved_bytes", "early_tiles", "bands", "banded_launches"):
            st[k] += st2[k]
        return st

    def _run(self, flags):
        _ffi.check(_ffi.lib().bt_preprocessor_run(self.pre._h, self.atlas._h, self.flags | flags))

    def _layout(self):
        if self._pieces is None:
            self._ranges = shard_ranges(self.pre)
            self._pieces = shard_pieces(self.pre)
            if self.result == "distributed" and self._pieces:  # the finest LOD stays on its owners
                finest = max(p["lod"] for p in self._pieces)
                self.held = [p for p in self._pieces if p["lod"] == finest and p["owner_rank"] == self.rank]
                self._ranges = [r for r in self._ranges if r["lod"] != finest]
                self._pieces = [p for p in self._pieces if p["lod"] != finest]
            self.gather_bytes = sum(p["layers"] * self.tile_bytes for p in self._pieces)

    def step(self, profile: bool = False, gather: bool = True):
        """gather=False skips the collectives (timing of the kernels alone; the atlas is then incomplete)."""
        import torch

        p = _ffi.RUN_PROFILE if profile else 0
        if self._comm is not None:
            flags = self.flags | p | (0 if gather else _ffi.RUN_SHARD_LOCAL)
            _ffi.check(_ffi.lib().bt_preprocessor_run_sharded(self.pre._h, self.atlas._h, self._comm, flags))
            if not gather and self.world > 1:
                self._run(_ffi.RUN_SHARD_FINISH)
            self._layout()
            return
        self._run(_ffi.RUN_SHARD_LOCAL | p)
        self._layout()
        if gather:
            with torch.cuda.stream(self.stream):  # same queue as the kernels: ordered without host syncs
                if self._ranges:
                    all_gather_ranges(self.storage, self.tile_bytes, self._ranges, self.rank, self.world, self.dist)
                else:
                    broadcast_pieces(self.storage, self.tile_bytes, self._pieces, self.dist)
        self._run(_ffi.RUN_SHARD_FINISH)

    # ---- the overlapped step: begin_step() = local kernels + the exchange on its own queue; finish_step() = the finishing kernels
    # behind that exchange.  Between the two this rank's compute stream is free for the local phase of its OTHER job.
    def begin_step(self, profile: bool = False):
        import torch

        assert not self.pending
        p = _ffi.RUN_PROFILE if profile else 0
        if self._comm is not None:
            _ffi.check(_ffi.lib().bt_preprocessor_run_sharded(self.pre._h, self.atlas._h, self._comm, self.flags | p | _ffi.RUN_SHARD_OVERLAP))
            self._layout()
            self.pending = True
            return
        if self._side_stream is None:
            self._side_stream = torch.cuda.Stream(device=self.stream.device)
            self._local_done, self._exchange_done = torch.cuda.Event(), torch.cuda.Event()
        self._run((_ffi.RUN_SHARD_LOCAL if self.world > 1 else 0) | p)
        self._layout()
        self._local_done.record(self.stream)
        self._side_stream.wait_event(self._local_done)
        if self.world > 1:
            with torch.cuda.stream(self._side_stream):
                if self._ranges:
                    all_gather_ranges(self.storage, self.tile_bytes, self._ranges, self.rank, self.world, self.dist)
                else:
                    broadcast_pieces(self.storage, self.tile_bytes, self._pieces, self.dist)
        self._exchange_done.record(self._side_stream)
        self.pending = True

    def finish_step(self, profile: bool = False):
        assert self.pending
        p = _ffi.RUN_PROFILE if profile else 0
        if self._comm is not None:
            _ffi.check(_ffi.lib().bt_preprocessor_finish_sharded(self.pre._h, self.atlas._h, self._comm, self.flags | p))
        else:
            self.stream.wait_event(self._exchange_done)
            if self.world > 1:
                self._run(_ffi.RUN_SHARD_FINISH | p)
        self.pending = False

    def exchange(self):
        """Only the exchange of a step (no kernels): the collective-only leg of the bench.  The queue must have run once."""
        import torch

        self._layout()
        if self._comm is not None:
            _ffi.check(_ffi.lib().bt_preprocessor_run_sharded(self.pre._h, self.atlas._h, self._comm, self.flags | _ffi.RUN_SHARD_EXCHANGE))
            return
        with torch.cuda.stream(self.stream):
            if self._ranges:
                all_gather_ranges(self.storage, self.tile_bytes, self._ranges, self.rank, self.world, self.dist)
            else:
                broadcast_pieces(self.storage, self.tile_bytes, self._pieces, self.dist)

    def stats(self):
        return self.pre.stats()

    def profile(self):
        return self.pre.profile()

    def close(self):
        if self._comm is not None and self._owns_comm and device_open(getattr(self.atlas, "device", None)):
            _ffi.lib().bt_comm_destroy(self._comm)
        self._comm = None


class OverlappedSharded:
    """Two jobs of one rank (an atlas and a preprocessor each, one communicator): step k's exchange runs on the collectives' queue
    while step k + 1's local kernels run on the compute stream; a step's finishing kernels follow its exchange one step late.
    Per step max(kernels, exchange) instead of their sum — what lets N > 1 scale when the gather (0.70 GB per rank for the 16k
    job) takes longer than a rank's share of the kernels."""

    def __init__(self, job_a: ShardedPreprocess, job_b: ShardedPreprocess):
        self.jobs = (job_a, job_b)
        self.k = 0

    def step(self, profile: bool = False):
        cur, prev = self.jobs[self.k & 1], self.jobs[(self.k + 1) & 1]
        if cur.pending:  # (only when the caller mixed step() and flush() oddly: finish before reusing the atlas)
            cur.finish_step()
        cur.begin_step(profile)
        if prev.pending:
            prev.finish_step(profile)
        self.k += 1

    def flush(self):
        for j in self.jobs:
            if j.pending:
                j.finish_step()
