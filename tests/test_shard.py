"""Multi-GPU sharding.  CPU part: the in-place all-gather layout over a gloo group of 2 processes (the
x-major atlas order makes each rank's strip one contiguous run of layers).  GPU part: G ranks simulated
one after the other on a single device must reproduce the oracle bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

import _cases as K
import _oracle as O


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _strip_ranges(tiles, lods, world):
    """(first_layer, layers_per_rank) per LOD from the atlas-index contract alone."""
    index = {c: i for c, i in tiles}
    out = []
    for lod in lods:
        n = 1 << lod
        out.append(dict(lod=lod, first_layer=index[(0, lod, 0, 0)], layers_per_rank=n // world * n))
    return out


def _gloo_worker(rank, world, port, tile_bytes, layers, ranges, full, owned_mask, result_queue):
    import torch
    import torch.distributed as dist

    from bevy_terrain_amd.shard import all_gather_ranges

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    storage = torch.zeros(layers * tile_bytes, dtype=torch.uint8)
    full_t = torch.from_numpy(full.reshape(-1))
    for layer in range(layers):  # this rank only holds the tiles it computed
        if owned_mask[rank][layer]:
            storage[layer * tile_bytes:(layer + 1) * tile_bytes] = full_t[layer * tile_bytes:(layer + 1) * tile_bytes]
    all_gather_ranges(storage, tile_bytes, ranges, rank, world, dist)
    gathered = np.zeros(layers, bool)
    for r in ranges:
        gathered[r["first_layer"]:r["first_layer"] + world * r["layers_per_rank"]] = True
    ok = all(torch.equal(storage[l * tile_bytes:(l + 1) * tile_bytes], full_t[l * tile_bytes:(l + 1) * tile_bytes])
             for l in range(layers) if gathered[l])
    untouched = all(bool((storage[l * tile_bytes:(l + 1) * tile_bytes] == 0).all()) for l in range(layers)
                    if not gathered[l] and not owned_mask[rank][l])
    result_queue.put((rank, ok, untouched))
    dist.destroy_process_group()


def test_inplace_allgather_layout_gloo_world2():
    import torch.multiprocessing as mp

    T, b, lod_count, world = 16, 2, 4, 2
    src = K.random_raster(O.FORMAT_R16, 120, 120, seed=5)
    oracle = K.oracle_planar(src, lod_count, T, b, O.FORMAT_R16, threads=2)
    tiles = oracle.tiles()
    layers = len(tiles)
    full = np.stack([oracle.tile(0, i) for _, i in tiles]).view(np.uint8)
    tile_bytes = T * T * 2
    lods = [3, 2, 1]  # the LODs the fused main kernel shards (finest three)
    ranges = _strip_ranges(tiles, lods, world)
    # contract check: layers [first + r*per, first + (r+1)*per) are exactly rank r's columns, x-major
    owned = [np.zeros(layers, bool) for _ in range(world)]
    for r in ranges:
        n = 1 << r["lod"]
        for rank in range(world):
            lo = r["first_layer"] + rank * r["layers_per_rank"]
            for k in range(r["layers_per_rank"]):
                coord = tiles[lo + k][0]
                assert coord == (0, r["lod"], rank * (n // world) + k // n, k % n)
                owned[rank][lo + k] = True
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(rank, world, port, tile_bytes, layers, ranges, full, owned, q))
             for rank in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True, True), (1, True, True)]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_ranks_simulated_on_one_device_match_oracle(world):
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_ranges

    device = bt.Device(0)
    T, b, lod_count = 64, 2, 6
    src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=31, holes=0.01)
    cfg = bt.TerrainConfig(lod_count=lod_count, atlas_size=2048, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, lod_count)),
                                                bt.AssetServer().insert("s", src), atlas)
    L = _ffi.lib()
    for rank in range(world):  # every "rank" writes its strip into the same atlas: no collective needed here
        _ffi.check(L.bt_preprocessor_set_shard(pre._h, rank, world))
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL))
        ranges = shard_ranges(pre)
        assert [(r["lod"], r["layers_per_rank"]) for r in ranges] == [(5, 32 // world * 32), (4, 16 // world * 16), (3, 8 // world * 8)]
        assert [r["first_layer"] for r in ranges] == [0, 1024, 1024 + 256]
    _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_FINISH))
    device.synchronize()
    oracle = K.oracle_planar(src, lod_count, T, b, O.FORMAT_R16, atlas_size=2048)
    assert K.assert_atlas_equal(atlas, oracle) == 1365


@pytest.mark.gpu
def test_unshardable_job_runs_everything_on_every_rank():
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_ranges

    device = bt.Device(0)
    src = K.random_raster(O.FORMAT_R16, 200, 200, seed=8)
    cfg = bt.TerrainConfig(lod_count=3, atlas_size=64, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=32, border_size=2))
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, 3)),
                                                bt.AssetServer().insert("s", src), atlas)
    _ffi.check(_ffi.lib().bt_preprocessor_set_shard(pre._h, 1, 8))  # the third-finest LOD has 1 column < 8 ranks
    _ffi.check(_ffi.lib().bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL | _ffi.RUN_SHARD_FINISH))
    assert shard_ranges(pre) == []
    device.synchronize()
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, 3, 32, 2, O.FORMAT_R16, atlas_size=64)) == 21


@pytest.mark.gpu
def test_nccl_single_rank_group_runs_the_sharded_step():
    """The 8-GPU run belongs to the driver; on the 1-GPU box at least the exact code path of bench.py --gpus N
    (device bytes -> torch tensor, in-place all_gather_into_tensor on the kernels' stream, LOCAL / FINISH runs)
    must work with a world of one RCCL rank."""
    import torch
    import torch.distributed as dist

    import bevy_terrain_amd as bt
    from bevy_terrain_amd.shard import ShardedPreprocess, all_gather_ranges

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        device = bt.Device(0)
        src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=33)
        cfg = bt.TerrainConfig(lod_count=4, atlas_size=128, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=128, border_size=2))
        atlas = bt.TileAtlas.new(cfg, device)
        job = ShardedPreprocess(bt.Preprocessor.new(), atlas, bt.AssetServer().insert("s", src), "s", range(0, 4), 0, 1)
        job.step()
        job.step(profile=True)
        # an explicit in-place gather over the atlas bytes (world 1: send == recv)
        with torch.cuda.stream(device.torch_stream):
            all_gather_ranges(job.storage, job.tile_bytes, [dict(first_layer=0, layers_per_rank=64)], 0, 1, dist)
        dist.barrier()
        torch.cuda.synchronize()
        assert K.assert_atlas_equal(atlas, K.oracle_planar(src, 4, 128, 2, O.FORMAT_R16, atlas_size=128)) == 85
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
def test_bench_two_ranks_end_to_end_on_one_gpu():
    """bench.py --gpus 2 as the driver launches it (torch.distributed.run, one process per rank), except that both
    ranks share GPU 0 and the collective is gloo (RCCL refuses two ranks on one device): every rank preprocesses its
    column strip of the 16k job, the in-place all-gathers assemble the atlas, and rank 0's atlas must equal the
    oracle's for all 1365 tiles."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BT_BENCH_BACKEND="gloo", BT_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--spinup-ms", "20", "--verify"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["collective_backend"] == "gloo"
    assert line["verify_vs_oracle"] == {"tiles": 1365, "identical": 1365, "index_contract": True}
    assert "cpu_baseline" not in line
