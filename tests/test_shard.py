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
@pytest.mark.parametrize("result,held_tiles", [("replicated", 1365), ("distributed", 512 + 341)])
def test_bench_two_ranks_end_to_end_on_one_gpu(result, held_tiles):
    """`python bench.py --gpus 2` — the plain command, which re-executes itself under torch.distributed.run (one process per
    rank) — except that both ranks share GPU 0 and the collective is gloo (RCCL refuses two ranks on one device): every rank preprocesses its
    column strip of the 16k job, the in-place all-gathers assemble the atlas, and rank 0's atlas must equal the
    oracle's for all 1365 tiles."""
    import json
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env.update(BT_BENCH_BACKEND="gloo", BT_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    # the PLAIN command (no torch.distributed.run in front): bench.py starts its two ranks itself
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--spinup-ms", "20", "--verify", "--result", result]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["collective_backend"] == "gloo"
    assert line["config"]["ranks_seen"] == 2 and line["config"]["jobs_in_flight"] == 1
    # replicated: rank 0 ends with all 1365 tiles; distributed: with its half of the 1024 finest tiles + the 341 below
    assert line["verify_vs_oracle"] == {"tiles": held_tiles, "identical": held_tiles, "index_contract": True}
    assert line["config"]["result"].startswith(result)
    assert "cpu_baseline" not in line
    # the extras of the same run: kernels alone, the exchange alone, the other result mode
    cfg = line["config"]
    assert cfg["kernels_only_ms_per_step"] > 0 and cfg["collective_only_ms_per_step"] > 0
    other = cfg["other_result_mode"]
    assert other["result"] == ("distributed" if result == "replicated" else "replicated") and other["ms_per_step"] > 0
    # ... and the overlapped pair (two atlases per rank, the exchange of a step behind the kernels of the next): timed, and the
    # second atlas verified like the first
    assert cfg[f"{result}_overlapped"]["ms_per_step"] > 0 and cfg[f"{result}_overlapped"]["atlases_per_rank"] == 2
    assert line["verify_vs_oracle_overlapped_second_atlas"] == {"tiles": held_tiles, "identical": held_tiles, "index_contract": True}
    full, quarter = 1024 * 524288 + 256 * 524288 + 64 * 524288, 256 * 524288 + 64 * 524288
    assert cfg["all_gather_bytes_per_rank"] == (full if result == "replicated" else quarter)
    # round 5: the one-tile-per-rank collective that ran before anything was timed, the N = 1 step of the same invocation, and the
    # scaling of every mode against it
    assert cfg["rccl_preflight"]["ranks"] == 2 and cfg["rccl_preflight"]["slot_bytes"] == 512 * 512 * 2 and "gloo" in cfg["rccl_preflight"]["through"]
    assert cfg["n1_same_invocation"]["ms_per_step"] > 0
    assert cfg["independent_jobs"]["jobs"] == 2 and cfg["independent_jobs"]["tiles_per_s"] > 0
    sc = cfg["scaling_vs_n1_same_invocation"]
    assert sc["headline"]["speedup"] > 0 and abs(sc["headline"]["efficiency"] - sc["headline"]["speedup"] / 2) < 1e-12
    assert f"{result}_overlapped" in sc and "kernels_only" in sc
    assert other["all_gather_bytes_per_rank"] == (quarter if result == "replicated" else full)
    # round 6: the end-to-end span with every rank a PCIe link (distributed result, streamed): the two ranks together wrote the reference's
    # directory once, each uploaded about half of the source and saved its half of the finest tiles + every second lower tile
    e2e = line["end_to_end_sharded"]
    assert e2e["files"] == 1365 and e2e["config_tc"] and e2e["ms"] > 0 and e2e["n1_same_invocation"]["ms"] > 0
    assert len(e2e["per_rank"]) == 2 and all(r["streamed"] and r["early_tiles"] == 512 for r in e2e["per_rank"])
    assert sum(r["saved_bytes"] for r in e2e["per_rank"]) == 1365 * 512 * 512 * 2
    assert all(0.5 * 2**29 <= r["uploaded_bytes"] < 0.56 * 2**29 for r in e2e["per_rank"])


def _unit_pieces(tiles, sides, lod_hi, world):
    """The exchange layout from the atlas-index contract alone: units = sides x strips (strip = one column of LOD
    lod_hi - 2), units / world consecutive units per rank, one piece per maximal run of strips with one owner."""
    index = {c: i for c, i in tiles}
    strips = 1 << (lod_hi - 2)
    per_rank = sides * strips // world
    pieces = []
    for side in range(sides):
        for lod in (lod_hi, lod_hi - 1, lod_hi - 2):
            n = 1 << lod
            cols = n // strips
            strip = 0
            while strip < strips:
                owner = (side * strips + strip) // per_rank
                end = strip + 1
                while end < strips and (side * strips + end) // per_rank == owner:
                    end += 1
                pieces.append(dict(side=side, lod=lod, first_layer=index[(side, lod, strip * cols, 0)], layers=(end - strip) * cols * n, owner_rank=owner))
                strip = end
    return pieces


def _gloo_broadcast_worker(rank, world, port, tile_bytes, layers, pieces, full, result_queue):
    import torch
    import torch.distributed as dist

    from bevy_terrain_amd.shard import broadcast_pieces

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    storage = torch.zeros(layers * tile_bytes, dtype=torch.uint8)
    full_t = torch.from_numpy(full.reshape(-1))
    covered = np.zeros(layers, bool)
    for p in pieces:
        lo, hi = p["first_layer"] * tile_bytes, (p["first_layer"] + p["layers"]) * tile_bytes
        covered[p["first_layer"]:p["first_layer"] + p["layers"]] = True
        if p["owner_rank"] == rank:  # this rank only holds what it computed
            storage[lo:hi] = full_t[lo:hi]
    broadcast_pieces(storage, tile_bytes, pieces, dist)
    ok = all(torch.equal(storage[l * tile_bytes:(l + 1) * tile_bytes], full_t[l * tile_bytes:(l + 1) * tile_bytes]) for l in range(layers) if covered[l])
    untouched = all(bool((storage[l * tile_bytes:(l + 1) * tile_bytes] == 0).all()) for l in range(layers) if not covered[l])
    result_queue.put((rank, ok, untouched))
    dist.destroy_process_group()


def test_cube_piece_layout_and_broadcast_gloo_world2():
    """The cube job's exchange (24 units, irregular: a rank's units span faces) as in-place broadcasts over a gloo
    group of 2 processes; the layout is derived here from the atlas-index contract (oracle queue), not from the library."""
    import torch.multiprocessing as mp

    T, b, lod_count, world = 8, 2, 4, 2
    faces = [K.random_raster(O.FORMAT_R16, 40, 40, seed=80 + s) for s in range(6)]
    oracle = O.OracleAtlas(lod_count, 6 * 85, True, [(T, b, 1, O.FORMAT_R16)])
    oracle.preprocess_spherical(0, faces, (0, lod_count)).run(2)
    tiles = oracle.tiles()
    layers = len(tiles)
    full = np.stack([oracle.tile(0, i) for _, i in tiles]).view(np.uint8)
    pieces = _unit_pieces(tiles, 6, lod_count - 1, world)
    # every tile of the three finest LODs is in exactly one piece, and a piece's layers are its owner's columns
    seen = np.zeros(layers, int)
    for p in pieces:
        seen[p["first_layer"]:p["first_layer"] + p["layers"]] += 1
        n, strips = 1 << p["lod"], 1 << (lod_count - 3)
        for k in range(p["layers"]):
            side, lod, x, y = tiles[p["first_layer"] + k][0]
            assert (side, lod) == (p["side"], p["lod"])
            assert (side * strips + x // (n // strips)) // (6 * strips // world) == p["owner_rank"]
    assert all(seen[i] == (1 if tiles[i][0][1] >= 1 else 0) for i in range(layers))
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_gloo_broadcast_worker, args=(rank, world, port, T * T * 2, layers, pieces, full, q)) for rank in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True, True), (1, True, True)]


def _emulate_ranks(device, world, make_job, n_tiles, oracle, lod_hi, T, b, sides=1):
    """A true multi-rank emulation on one device: every rank gets its OWN atlas and runs BT_RUN_SHARD_LOCAL; rank 0's
    atlas then receives every piece from its owner's atlas (what the in-place broadcast does) and runs
    BT_RUN_SHARD_FINISH.  Before the exchange a rank's pieces must already hold finished centres (and, at the finest
    LOD, finished tiles): nobody else computes them."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_pieces

    L = _ffi.lib()
    jobs = []
    for rank in range(world):
        atlas, pre = make_job()
        _ffi.check(L.bt_preprocessor_set_shard(pre._h, rank, world))
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL))
        jobs.append((atlas, pre))
    device.synchronize()
    pieces = shard_pieces(jobs[0][1])
    assert pieces and all(shard_pieces(pre) == pieces for _, pre in jobs)
    coords = {i: c for c, i in oracle.tiles()}
    c = T - 2 * b
    for p in pieces:
        data = jobs[p["owner_rank"]][0].download_tiles(0, p["first_layer"], p["layers"])
        for k in range(0, p["layers"], max(1, p["layers"] // 9)):
            exp = oracle.tile(0, p["first_layer"] + k)
            side, lod, x, y = coords[p["first_layer"] + k]
            on_face_edge = x in (0, (1 << lod) - 1) or y in (0, (1 << lod) - 1)
            if p["lod"] == lod_hi and not (sides == 6 and on_face_edge):  # (cube seams are stitched after the exchange)
                assert np.array_equal(data[k], exp), (p, coords[p["first_layer"] + k])
            else:
                assert np.array_equal(data[k][b:b + c, b:b + c], exp[b:b + c, b:b + c]), (p, coords[p["first_layer"] + k])
        if p["owner_rank"] != 0:  # the broadcast, by hand
            for k in range(p["layers"]):
                jobs[0][0].upload_tile(0, p["first_layer"] + k, data[k])
    atlas0, pre0 = jobs[0]
    _ffi.check(L.bt_preprocessor_run(pre0._h, atlas0._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_FINISH))
    device.synchronize()
    assert K.assert_atlas_equal(atlas0, oracle) == n_tiles
    return pieces


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_cube_job_sharded_over_emulated_ranks(world):
    """BASELINE config 5's shape (6 faces, lod_count 5 -> 2046 tiles) at T = 32: 24 units, 24 / world per rank."""
    import bevy_terrain_amd as bt

    device = bt.Device(0)
    T, b, lods, W = 32, 2, 5, 470
    faces = [K.random_raster(O.FORMAT_R16, W, W, seed=90 + s, holes=0.003) for s in range(6)]
    paths = [f"face{s}" for s in range(6)]

    def make_job():
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/spherical")
        cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer()
        for p, f in zip(paths, faces):
            server.insert(p, f)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
            bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
        return atlas, pre

    oracle = O.OracleAtlas(lods, 2048, True, [(T, b, 1, O.FORMAT_R16)])
    oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
    pieces = _emulate_ranks(device, world, make_job, 2046, oracle, lods - 1, T, b, sides=6)
    assert pieces == [dict(p, attachment_index=0) for p in _unit_pieces(oracle.tiles(), 6, lods - 1, world)]
    assert len({p["owner_rank"] for p in pieces}) == world


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 8])
def test_planar_job_sharded_over_emulated_ranks(world):
    import bevy_terrain_amd as bt

    device = bt.Device(0)
    T, b, lods = 64, 2, 6
    src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=35, holes=0.01)

    def make_job():
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, lods)), bt.AssetServer().insert("s", src), atlas)
        return atlas, pre

    oracle = K.oracle_planar(src, lods, T, b, O.FORMAT_R16, atlas_size=2048)
    pieces = _emulate_ranks(device, world, make_job, 1365, oracle, lods - 1, T, b)
    assert len(pieces) == 3 * world


@pytest.mark.gpu
def test_library_issued_collective_single_rank():
    """bt_preprocessor_run_sharded through the library's own RCCL communicator (bt_comm_unique_id / bt_comm_create): a
    world of one rank is all a 1-GPU box can form, but it is the code path bench.py --gpus N drives on 8."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd.shard import ShardedPreprocess

    device = bt.Device(0)
    src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=36)
    cfg = bt.TerrainConfig(lod_count=4, atlas_size=128, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=128, border_size=2))
    atlas = bt.TileAtlas.new(cfg, device)
    job = ShardedPreprocess(bt.Preprocessor.new(), atlas, bt.AssetServer().insert("s", src), "s", range(0, 4), 0, 1, collective="library")
    bt._ffi.check(bt._ffi.lib().bt_comm_check(job._comm))  # grouped ncclAllGather + ncclBroadcast through the communicator
    import ctypes
    ms = ctypes.c_float(-1.0)  # the tile-sized preflight bench.py --gpus N runs before its timed steps
    bt._ffi.check(bt._ffi.lib().bt_comm_preflight(job._comm, 512 * 512 * 2, ctypes.byref(ms)))
    assert ms.value >= 0.0
    assert bt._ffi.lib().bt_comm_preflight(job._comm, 0, None) != 0  # a slot of no bytes is refused
    job.step()
    job.step(profile=True)
    device.synchronize()
    job.close()
    assert K.assert_atlas_equal(atlas, K.oracle_planar(src, 4, 128, 2, O.FORMAT_R16, atlas_size=128)) == 85


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(14))
def test_random_jobs_sharded_over_emulated_ranks(seed):
    """Random tile shapes, LOD counts, formats, hole densities and world sizes (incl. 3 and 6 for the cube's 24 units):
    whatever the planner decides — strips, or "not shardable: every rank runs everything" — rank 0 ends with the
    oracle's atlas."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_pieces

    rng = np.random.default_rng(4_000 + seed)
    device = bt.Device(0)
    cube = seed % 3 == 2
    T = int(rng.choice([16, 24, 32, 64]))
    b = int(rng.choice([1, 2, 4]))
    lods = int(rng.integers(3, 6 if cube else 7))
    fmt = O.FORMAT_R16 if rng.random() < 0.75 else O.FORMAT_RGBA8
    world = int(rng.choice([2, 3, 4, 6, 8] if cube else [2, 4, 8]))
    holes = float(rng.choice([0.0, 0.01, 0.2]))
    W = int(((T - 2 * b) << (lods - 1)) * rng.uniform(0.4, 1.5))
    n_tiles = (6 if cube else 1) * sum(4 ** l for l in range(lods))
    size = n_tiles + 8
    if cube:
        faces = [K.random_raster(fmt, W, W, seed=300 + 10 * seed + s, holes=holes) for s in range(6)]
        paths = [f"face{s}" for s in range(6)]
        oracle = O.OracleAtlas(lods, size, True, [(T, b, 1, fmt)])
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
    else:
        src = K.random_raster(fmt, W, W + 7, seed=300 + seed, holes=holes)
        oracle = K.oracle_planar(src, lods, T, b, fmt, atlas_size=size, threads=O.usable_cores())

    def make_job():
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=size, path="terrains/sweep",
                               **({} if cube else dict(model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))))
        cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer()
        pre = bt.Preprocessor.new().clear_attachment(0, atlas)
        if cube:
            for p, f in zip(paths, faces):
                server.insert(p, f)
            pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas)
        else:
            pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="s", lod_range=range(0, lods)), server.insert("s", src), atlas)
        return atlas, pre

    probe_atlas, probe = make_job()
    L = _ffi.lib()
    _ffi.check(L.bt_preprocessor_set_shard(probe._h, world - 1, world))
    _ffi.check(L.bt_preprocessor_run(probe._h, probe_atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL))
    if shard_pieces(probe):
        pieces = _emulate_ranks(device, world, make_job, n_tiles, oracle, lods - 1, T, b, sides=6 if cube else 1)
        assert len({p["owner_rank"] for p in pieces}) == world
    else:  # not shardable: LOCAL + FINISH on any rank is the whole job
        _ffi.check(L.bt_preprocessor_run(probe._h, probe_atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_FINISH))
        device.synchronize()
        assert K.assert_atlas_equal(probe_atlas, oracle) == n_tiles


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_planar_job_distributed_result_over_emulated_ranks(world, tmp_path):
    """BT_RUN_SHARD_DISTRIBUTED: the finest LOD is not exchanged.  One atlas per emulated rank; only the two parent LODs
    travel (to every rank); afterwards rank r holds its own finest tiles and EVERY lower LOD, all equal to the oracle's,
    and the ranks' saves together write every tile file exactly once (+ config.tc from rank 0)."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_pieces

    device = bt.Device(0)
    L = _ffi.lib()
    T, b, lods = 64, 2, 6
    src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=36, holes=0.01)
    oracle = K.oracle_planar(src, lods, T, b, O.FORMAT_R16, atlas_size=2048)
    jobs = []
    D = _ffi.RUN_SHARD_DISTRIBUTED | _ffi.RUN_KEEP_QUEUE
    for rank in range(world):
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/dist", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, str(tmp_path)).preprocess_tile(
            bt.PreprocessDataset(path="s", lod_range=range(0, lods)), bt.AssetServer().insert("s", src), atlas)
        _ffi.check(L.bt_preprocessor_set_shard(pre._h, rank, world))
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, D | _ffi.RUN_SHARD_LOCAL))
        jobs.append((atlas, pre))
    device.synchronize()
    pieces = shard_pieces(jobs[0][1])
    finest = max(p["lod"] for p in pieces)
    assert finest == lods - 1 and len(pieces) == 3 * world
    moved = 0
    for p in pieces:
        if p["lod"] == finest:
            continue  # stays on its owner
        data = jobs[p["owner_rank"]][0].download_tiles(0, p["first_layer"], p["layers"])
        moved += data.nbytes
        for r, (atlas, _) in enumerate(jobs):
            if r != p["owner_rank"]:
                for k in range(p["layers"]):
                    atlas.upload_tile(0, p["first_layer"] + k, data[k])
    total = sum(p["layers"] for p in pieces) * T * T * 2
    assert moved * 4 < total * 1.3  # about a quarter of the replicated exchange
    for atlas, pre in jobs:
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, D | _ffi.RUN_SHARD_FINISH))
    device.synchronize()
    coords = {i: c for c, i in oracle.tiles()}
    for rank, (atlas, pre) in enumerate(jobs):
        held = [p for p in pieces if p["lod"] == finest and p["owner_rank"] == rank]
        assert held
        layers = set()
        for p in held:
            layers.update(range(p["first_layer"], p["first_layer"] + p["layers"]))
        layers.update(i for i, c in coords.items() if c[1] < finest)
        data = atlas.download_tiles(0, 0, 1365)
        bad = [coords[i] for i in sorted(layers) if not np.array_equal(data[i], oracle.tile(0, i))]
        assert not bad, (rank, len(bad), bad[:4])
        pre.save(atlas, str(tmp_path))
    directory = jobs[0][0].attachment_directory(str(tmp_path), 0)
    files = sorted(f for f in os.listdir(directory) if f.endswith(".bin"))
    assert len(files) == 1365
    import ctypes

    def name_of(c):
        buf = ctypes.create_string_buffer(64)
        L.bt_tile_name(bt.TileCoordinate(*c)._c(), buf, 64)
        return buf.value.decode()

    by_name = {name_of(c): i for i, c in coords.items()}
    for f in files[::23]:
        tile = np.fromfile(os.path.join(directory, f), dtype=np.uint16).reshape(T, T)
        assert np.array_equal(tile, oracle.tile(0, by_name[f[:-4]])), f
    assert os.path.exists(os.path.join(str(tmp_path), "terrains/dist", "config.tc"))


@pytest.mark.gpu
def test_distributed_result_is_refused_for_cube_jobs():
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi

    device = bt.Device(0)
    T, b, lods = 32, 2, 4
    faces = [K.random_raster(O.FORMAT_R16, 200, 200, seed=90 + s) for s in range(6)]
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=600, path="terrains/spherical")
    cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer()
    for s in range(6):
        server.insert(f"f{s}", faces[s])
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
        bt.SphericalDataset(attachment_index=0, paths=[f"f{s}" for s in range(6)], lod_range=range(0, lods)), server, atlas)
    _ffi.check(_ffi.lib().bt_preprocessor_set_shard(pre._h, 1, 4))
    rc = _ffi.lib().bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL | _ffi.RUN_SHARD_DISTRIBUTED)
    assert rc == -5  # BT_ERR_UNSUPPORTED


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_rgba8_job_at_tile_size_512_sharded_over_emulated_ranks(world):
    """An Rgba8 (albedo) job at the reference's tile shape through the sharded path: 16384^2 Rgba8, lod_count 6, T = 512 ->
    1365 tiles of 1 MiB (fused_direct), 8 units; every emulated rank has its own atlas and runs its strip, rank 0 receives
    the pieces and finishes; its atlas must equal the checker's (the reference's WGSL executed on the CPU)."""
    import bevy_terrain_amd as bt

    device = bt.Device(0)
    T, b, lods, W = 512, 2, 6, 16384
    rng = np.random.default_rng(77)
    coarse = rng.integers(1, 256, size=(W // 8, W // 8, 4), dtype=np.uint8)
    src = np.repeat(np.repeat(coarse, 8, axis=0), 8, axis=1)  # 1 GiB, blocky (cheap to make; every texel still filtered)
    src[..., 3] = 255
    src[5000:5200, 9000:9300, 0] = 0
    src = np.ascontiguousarray(src)
    ptr = device.upload(src)

    def make_job():
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=T, border_size=b, format=bt.AttachmentFormat.Rgba8))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, lods)), bt.AssetServer().insert("s", (ptr, W, W)), atlas)
        return atlas, pre

    oracle = K.reference_kernels(O.OracleAtlas(lods, 2048, False, [(T, b, 1, O.FORMAT_RGBA8)]))
    oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(O.usable_cores())
    pieces = _emulate_ranks(device, world, make_job, 1365, oracle, lods - 1, T, b)
    assert len({p["owner_rank"] for p in pieces}) == world
    device.free(ptr)


class _LocalGroup:
    """An in-process stand-in for torch.distributed over EMULATED ranks (one process, one GPU): a rank's collective calls are
    recorded; `run()` then moves the bytes between the ranks' atlases the way the collectives do (in-place all-gather: every
    rank's slice to every rank; broadcast: the owner's piece to every rank)."""

    def __init__(self, world):
        self.world, self.ops = world, {r: [] for r in range(world)}

    def for_rank(self, rank):
        group = self

        class Dist:
            def all_gather_into_tensor(self, whole, mine, group_=None, group=None):
                self_ops.append(("gather", whole, mine))

            def broadcast(self, tensor, src, group=None):
                self_ops.append(("broadcast", tensor, src))

        self_ops = group.ops[rank]
        return Dist()

    def run(self):
        import torch

        torch.cuda.synchronize()
        n = len(self.ops[0])
        assert all(len(v) == n for v in self.ops.values())
        for i in range(n):
            kind = self.ops[0][i][0]
            if kind == "gather":
                count = self.ops[0][i][2].numel()
                slices = [self.ops[r][i][2].clone() for r in range(self.world)]  # (in place: a rank's slice aliases its `whole`)
                for dst in range(self.world):
                    for src in range(self.world):
                        self.ops[dst][i][1][src * count:(src + 1) * count].copy_(slices[src])
            else:
                src = self.ops[0][i][2]
                data = self.ops[src][i][1].clone()
                for dst in range(self.world):
                    self.ops[dst][i][1].copy_(data)
        torch.cuda.synchronize()
        for v in self.ops.values():
            v.clear()


@pytest.mark.gpu
@pytest.mark.parametrize("world,cube", [(2, False), (4, False), (8, False), (4, True)])
def test_overlapped_steps_over_emulated_ranks(world, cube):
    """The overlapped step (begin_step: local kernels + the exchange on its own queue; finish_step: the finishing kernels behind
    it, one step late; TWO atlases per rank that alternate) over emulated ranks: after four steps and the flush BOTH atlases of
    EVERY rank equal the oracle's.  The ranks' collectives are the in-process stand-in above; the real ones run in
    test_bench_two_ranks_end_to_end_on_one_gpu."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd.shard import ShardedPreprocess

    device = bt.Device(0)
    T, b = (32, 2) if cube else (64, 2)
    lods = 4 if cube else 6
    if cube:
        faces = [K.random_raster(O.FORMAT_R16, 230, 230, seed=60 + s, holes=0.003) for s in range(6)]
        paths = [f"face{s}" for s in range(6)]
        oracle = O.OracleAtlas(lods, 1024, True, [(T, b, 1, O.FORMAT_R16)])
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
        n_tiles = 6 * 85
    else:
        src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=37, holes=0.01)
        oracle = K.oracle_planar(src, lods, T, b, O.FORMAT_R16, atlas_size=2048)
        n_tiles = 1365
    group = _LocalGroup(world)
    jobs = []  # [rank][slot]
    for rank in range(world):
        pair = []
        for _ in range(2):
            if cube:
                cfg = bt.TerrainConfig(lod_count=lods, atlas_size=1024, path="terrains/spherical")
                cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16))
                server = bt.AssetServer()
                for p, f in zip(paths, faces):
                    server.insert(p, f)
                path = paths
            else:
                cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
                cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
                server, path = bt.AssetServer().insert("s", src), "s"
            atlas = bt.TileAtlas.new(cfg, device)
            pair.append(ShardedPreprocess(bt.Preprocessor.new().clear_attachment(0, atlas), atlas, server, path, range(0, lods), rank, world,
                                          collective="torch", dist=group.for_rank(rank)))
        jobs.append(pair)
    for k in range(4):
        cur, prev = k & 1, (k + 1) & 1
        for rank in range(world):
            jobs[rank][cur].begin_step()
        group.run()  # step k's exchange ...
        for rank in range(world):  # ... and, behind it on the compute streams, step k - 1's finishing kernels
            if jobs[rank][prev].pending:
                jobs[rank][prev].finish_step()
    for rank in range(world):
        for j in jobs[rank]:
            if j.pending:
                j.finish_step()
    device.synchronize()
    for rank in range(world):
        for j in jobs[rank]:
            assert K.assert_atlas_equal(j.atlas, oracle) == n_tiles, rank


@pytest.mark.gpu
def test_library_overlapped_step_single_rank():
    """BT_RUN_SHARD_OVERLAP + bt_preprocessor_finish_sharded through the library's own communicator and its collectives' stream:
    a world of one rank (all a 1-GPU box can form), two atlases alternating under OverlappedSharded."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd.shard import OverlappedSharded, ShardedPreprocess

    device = bt.Device(0)
    src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=38)
    oracle = K.oracle_planar(src, 4, 128, 2, O.FORMAT_R16, atlas_size=128)

    def make(comm=None):
        cfg = bt.TerrainConfig(lod_count=4, atlas_size=128, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=128, border_size=2))
        atlas = bt.TileAtlas.new(cfg, device)
        return ShardedPreprocess(bt.Preprocessor.new(), atlas, bt.AssetServer().insert("s", src), "s", range(0, 4), 0, 1, collective="library", comm=comm)

    a = make()
    b = make(comm=a._comm)
    pair = OverlappedSharded(a, b)
    for k in range(5):
        pair.step(profile=k == 3)
    with pytest.raises(bt._ffi.BtError):  # a pending step must be finished before the same preprocessor starts another
        pending = a if a.pending else b
        bt._ffi.check(bt._ffi.lib().bt_preprocessor_run_sharded(pending.pre._h, pending.atlas._h, pending._comm, pending.flags | bt._ffi.RUN_SHARD_OVERLAP))
    pair.flush()
    device.synchronize()
    for j in (a, b):
        assert K.assert_atlas_equal(j.atlas, oracle) == 85
    b.close()
    a.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,cube", [(2, False), (8, False), (4, True)])
def test_a_rank_reads_only_its_window_of_the_source(world, cube):
    """a rank needs its column strips of the source + the halo, nothing else.  Every emulated rank gets a host raster
    that is GARBAGE (zeros = no data) outside the window bt_preprocessor_source_window reports and hands it over deferred: the
    library uploads only the window (bytes asserted), and the rank's pieces must still come out as the oracle computes them
    from the intact raster."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_pieces

    device = bt.Device(0)
    L = _ffi.lib()
    if cube:
        T, b, lods, W = 32, 2, 4, 230
        faces = [K.random_raster(O.FORMAT_R16, W, W, seed=70 + s) for s in range(6)]
        oracle = O.OracleAtlas(lods, 1024, True, [(T, b, 1, O.FORMAT_R16)])
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(O.usable_cores())
    else:
        T, b, lods, W = 64, 2, 6, 2100
        faces = [K.random_raster(O.FORMAT_R16, W, W, seed=39)]
        oracle = K.oracle_planar(faces[0], lods, T, b, O.FORMAT_R16, atlas_size=2048)
    c = T - 2 * b
    total_uploaded = 0
    for rank in range(world):
        def job(rasters):
            if cube:
                cfg = bt.TerrainConfig(lod_count=lods, atlas_size=1024, path="terrains/spherical")
                cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16))
                atlas = bt.TileAtlas.new(cfg, device)
                server = bt.AssetServer()
                paths = [f"f{s}" for s in range(6)]
                for p, f in zip(paths, rasters):
                    server.insert(p, f)
                pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_spherical(
                    bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas, defer_upload=True)
            else:
                cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
                cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
                atlas = bt.TileAtlas.new(cfg, device)
                pre = bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, lods)), bt.AssetServer().insert("s", rasters[0]), atlas,
                                                            defer_upload=True)
            _ffi.check(L.bt_preprocessor_set_shard(pre._h, rank, world))
            return atlas, pre

        # first the windows (the plan is compiled, nothing travels yet) ...
        atlas, pre = job(faces)
        windows = [pre.source_window(atlas, i)[0] for i in range(len(faces))]
        # ... then the real run from rasters that hold nothing outside them
        poisoned = []
        for f, (x0, y0, x1, y1) in zip(faces, windows):
            g = np.zeros_like(f)
            g[y0:y1, x0:x1] = f[y0:y1, x0:x1]
            poisoned.append(g)
        atlas, pre = job(poisoned)
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL))
        device.synchronize()
        uploaded = pre.source_window(atlas, len(faces) - 1)[1]
        x0, y0, x1, y1 = windows[-1]
        assert uploaded == (x1 - x0) * (y1 - y0) * 2
        area = sum((w[2] - w[0]) * (w[3] - w[1]) for w in windows)
        total_uploaded += area * 2
        # a rank's share of the source: 1 / world of it + the halo of its strips (a strip of 8-texel-aligned staging windows)
        assert area * world < 1.35 * sum(f.size for f in faces), (rank, windows)
        if not cube:
            assert (y0, y1) == (0, W) and x1 - x0 < W // world + 2 * (T + 16)
        pieces = [p for p in shard_pieces(pre) if p["owner_rank"] == rank]
        assert pieces
        coords = {i: cc for cc, i in oracle.tiles()}
        finest = max(p["lod"] for p in pieces)
        for p in pieces:
            data = atlas.download_tiles(0, p["first_layer"], p["layers"])
            for k in range(0, p["layers"], max(1, p["layers"] // 11)):
                exp = oracle.tile(0, p["first_layer"] + k)
                side, lod, x, y = coords[p["first_layer"] + k]
                edge = x in (0, (1 << lod) - 1) or y in (0, (1 << lod) - 1)
                if p["lod"] == finest and not (cube and edge):
                    assert np.array_equal(data[k], exp), (rank, p, coords[p["first_layer"] + k])
                else:
                    assert np.array_equal(data[k][b:b + c, b:b + c], exp[b:b + c, b:b + c]), (rank, p)
    assert total_uploaded < 1.35 * sum(f.nbytes for f in faces)


@pytest.mark.gpu
def test_two_attachments_sharded_from_deferred_rasters_get_a_window_each():
    """the source window is decided PER RASTER.  One queue with a height job (R16: fused_main) and an albedo job
    (Rgba8: fused_direct), both rasters handed over deferred, two emulated ranks.  Each rank's two windows are its strips + halo of
    THAT raster (round 4 reported an empty window for the raster only fused_direct reads and uploaded nothing for it); the ranks run
    from rasters that hold nothing outside their windows, the pieces of both attachments are exchanged by hand, and after
    BT_RUN_SHARD_FINISH rank 0 holds the oracle's tiles of both attachments.  A later BT_RUN_GENERIC run of the kept queue needs the whole
    rasters: the missing part travels then (the uploaded window is remembered per raster)."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_pieces

    device = bt.Device(0)
    L = _ffi.lib()
    world, T, b, lods, W = 2, 64, 2, 5, 1000
    height = K.random_raster(O.FORMAT_R16, W, W, seed=81, holes=0.01)
    albedo = K.random_raster(O.FORMAT_RGBA8, W, W, seed=82, holes=0.01)
    n_tiles = sum(4 ** l for l in range(lods))
    oracle = O.OracleAtlas(lods, 512, False, [(T, b, 1, O.FORMAT_R16), (T, b, 1, O.FORMAT_RGBA8)])
    oracle.clear_attachment(0).preprocess_tile(0, height, (0, lods)).clear_attachment(1).preprocess_tile(1, albedo, (0, lods)).run(O.usable_cores())

    def job(rank, h, a):
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=512, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16))
        cfg.add_attachment(bt.AttachmentConfig(name="albedo", texture_size=T, border_size=b, format=bt.AttachmentFormat.Rgba8))
        atlas = bt.TileAtlas.new(cfg, device)
        server = bt.AssetServer().insert("h", h).insert("a", a)
        pre = bt.Preprocessor.new().clear_attachment(0, atlas).clear_attachment(1, atlas)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="h", lod_range=range(0, lods)), server, atlas, defer_upload=True)
        pre.preprocess_tile(bt.PreprocessDataset(attachment_index=1, path="a", lod_range=range(0, lods)), server, atlas, defer_upload=True)
        _ffi.check(L.bt_preprocessor_set_shard(pre._h, rank, world))
        return atlas, pre

    jobs = []
    for rank in range(world):
        atlas, pre = job(rank, height, albedo)
        windows = [pre.source_window(atlas, i)[0] for i in range(2)]
        for (x0, y0, x1, y1) in windows:  # a strip of the raster: every row, about half the columns + the halo
            assert (y0, y1) == (0, W) and 0 < x1 - x0 < W // world + 2 * (T + 16), windows
        poisoned = []
        for f, (x0, y0, x1, y1) in zip((height, albedo), windows):
            g = np.zeros_like(f)
            g[y0:y1, x0:x1] = f[y0:y1, x0:x1]
            poisoned.append(g)
        atlas, pre = job(rank, *poisoned)
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL))
        device.synchronize()
        x0, y0, x1, y1 = windows[1]
        assert pre.source_window(atlas, 1)[1] == (x1 - x0) * (y1 - y0) * 4  # the albedo window travelled, not nothing and not everything
        jobs.append((atlas, pre, windows, poisoned))
    pieces = shard_pieces(jobs[0][1])
    assert {p["attachment_index"] for p in pieces} == {0, 1} and {p["owner_rank"] for p in pieces} == {0, 1}
    for p in pieces:  # the exchange, by hand
        if p["owner_rank"] != 0:
            data = jobs[p["owner_rank"]][0].download_tiles(p["attachment_index"], p["first_layer"], p["layers"])
            for k in range(p["layers"]):
                jobs[0][0].upload_tile(p["attachment_index"], p["first_layer"] + k, data[k])
    atlas0, pre0, _, _ = jobs[0]
    _ffi.check(L.bt_preprocessor_run(pre0._h, atlas0._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_FINISH))
    device.synchronize()
    assert K.assert_atlas_equal(atlas0, oracle, 0) == n_tiles
    assert K.assert_atlas_equal(atlas0, oracle, 1) == n_tiles

    # the kept queue once more, unsharded and through the generic plan: its launches read the WHOLE rasters — the part that never
    # travelled is fetched from the caller's rows (here: the intact ones, handed over again under the same pointers' lifetime rule)
    atlas1, pre1 = job(1, height, albedo)
    _ffi.check(L.bt_preprocessor_run(pre1._h, atlas1._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL))
    device.synchronize()
    _ffi.check(L.bt_preprocessor_set_shard(pre1._h, 0, 1))
    _ffi.check(L.bt_preprocessor_run(pre1._h, atlas1._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_GENERIC))
    device.synchronize()
    assert pre1.source_window(atlas1, 1, generic=True)[1] == albedo.nbytes  # (the last raster that travelled: all of the albedo, not rank 1's strip)
    assert K.assert_atlas_equal(atlas1, oracle, 0) == n_tiles
    assert K.assert_atlas_equal(atlas1, oracle, 1) == n_tiles


@pytest.mark.gpu
def test_profile_of_a_sharded_step_counts_each_launch_once():
    """BT_RUN_SHARD_LOCAL and BT_RUN_SHARD_FINISH of one step are two profiled runs that each leave a whole event
    row; a launch is averaged over the rows that executed it (round 4 divided by both and halved every time)."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi

    device = bt.Device(0)
    L = _ffi.lib()
    src = K.random_raster(O.FORMAT_R16, 1100, 1100, seed=5)
    cfg = bt.TerrainConfig(lod_count=5, atlas_size=512, path="t", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=64, border_size=2))
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, 5)), bt.AssetServer().insert("s", src), atlas)
    _ffi.check(L.bt_preprocessor_set_shard(pre._h, 0, 2))
    steps = 3
    for _ in range(steps):
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_LOCAL | _ffi.RUN_PROFILE))
        _ffi.check(L.bt_preprocessor_run(pre._h, atlas._h, _ffi.RUN_KEEP_QUEUE | _ffi.RUN_SHARD_FINISH | _ffi.RUN_PROFILE))
    prof = pre.profile()
    assert len(prof) >= 3 and prof[0]["kind"] == "fused_main"
    assert all(l["samples"] == steps and l["avg_ms"] > 0 for l in prof), prof


@pytest.mark.gpu
@pytest.mark.parametrize("world,T,W,lods", [(2, 64, 1100, 6), (4, 64, 1100, 6), (8, 64, 1100, 6), (2, 512, 4096, 4)])
def test_sharded_streamed_distributed_end_to_end_over_emulated_ranks(world, T, W, lods, tmp_path):
    """bt_preprocessor_run_streamed_sharded with a host-side exchange (comm = NULL): every emulated rank (its own atlas) uploads only its
    window of the deferred source band by band, runs its units and writes its finest tiles while later bands run (BT_RUN_SHARD_LOCAL);
    the two parent LODs are moved between the atlases by the test; the finishing call (BT_RUN_SHARD_FINISH) runs the finishing kernels and
    writes the rank's share of the lower LODs.  Together the ranks write every tile file exactly once, byte-identical to the oracle's
    tiles, + config.tc from rank 0; a rank uploads about 1 / world of the source and saves about 1 / world of the bytes."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi
    from bevy_terrain_amd.shard import shard_pieces

    device = bt.Device(0)
    L = _ffi.lib()
    b = 2
    n_tiles = sum(4 ** l for l in range(lods))
    src = K.random_raster(O.FORMAT_R16, W, W, seed=36 + world, holes=0.01)
    src[W // 2 - 9:W // 2 + 9, :] = 0  # no data across every rank's strip (and a band seam)
    oracle = K.oracle_planar(src, lods, T, b, O.FORMAT_R16, atlas_size=2048)
    root = str(tmp_path)
    jobs, stats = [], []
    for rank in range(world):
        cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2048, path="terrains/dist", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
        cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new()
        if rank == 0:
            pre.clear_attachment(0, atlas, root)
        pre.preprocess_tile(bt.PreprocessDataset(path="s", lod_range=range(0, lods)), bt.AssetServer().insert("s", src), atlas, defer_upload=True)
        _ffi.check(L.bt_preprocessor_set_shard(pre._h, rank, world))
        jobs.append((atlas, pre))
    directory = jobs[0][0].attachment_directory(root, 0)
    seen = set()
    for rank, (atlas, pre) in enumerate(jobs):
        st = pre.run_streamed_sharded(atlas, root, local=True, finish=False)
        assert st["streamed"] and st["banded_launches"] == 1 and st["bands"] >= 2, st
        assert st["early_tiles"] == 4 ** (lods - 1) // world and st["saved_bytes"] == st["early_tiles"] * T * T * 2
        assert st["uploaded_bytes"] < 1.35 * src.nbytes / world + 64 * W * 2, (st, src.nbytes)
        assert pre.stats()["prev_zero_launches"] == 1
        now = set(os.listdir(directory))
        assert len(now - seen) == st["early_tiles"]  # this rank's finest files, none of them written before
        seen = now
        stats.append(st)
    pieces = shard_pieces(jobs[0][1])
    finest = max(p["lod"] for p in pieces)
    for p in pieces:
        if p["lod"] == finest:
            continue  # stays on its owner
        data = jobs[p["owner_rank"]][0].download_tiles(0, p["first_layer"], p["layers"])
        for r, (atlas, _) in enumerate(jobs):
            if r != p["owner_rank"]:
                for k in range(p["layers"]):
                    atlas.upload_tile(0, p["first_layer"] + k, data[k])
    total_saved = 0
    for rank, (atlas, pre) in enumerate(jobs):
        st = pre.run_streamed_sharded(atlas, root, local=False, finish=True)
        now = set(os.listdir(directory))
        total_saved += len(now - seen)
        seen = now
    assert len(seen) == n_tiles and total_saved == n_tiles - 4 ** (lods - 1)
    assert os.path.exists(os.path.join(root, "terrains/dist", "config.tc"))
    for c, i in oracle.tiles():
        tile = np.fromfile(os.path.join(directory, f"{c[0]}_{c[1]}_{c[2]}_{c[3]}.bin"), dtype=np.uint16).reshape(T, T)
        assert np.array_equal(tile, oracle.tile(0, i)), c
    assert sum(s["uploaded_bytes"] for s in stats) < 1.35 * src.nbytes + world * 64 * W * 2


@pytest.mark.gpu
def test_sharded_streamed_with_the_library_communicator_single_rank(tmp_path):
    """the one-call form (communicator given, both halves): a world of one through bt_preprocessor_run_streamed_sharded behaves like
    bt_preprocessor_run_streamed; a sharded preprocessor is refused by the unsharded entry point, cube jobs by the sharded one."""
    import bevy_terrain_amd as bt
    from bevy_terrain_amd import _ffi

    device = bt.Device(0)
    L = _ffi.lib()
    T, b, lods, W = 64, 2, 5, 700
    src = K.random_raster(O.FORMAT_R16, W, W, seed=77, holes=0.02)
    oracle = K.oracle_planar(src, lods, T, b, O.FORMAT_R16, atlas_size=512)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=512, path="terrains/one", model=bt.TerrainModel.planar((0, 0, 0), 1.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="h", texture_size=T, border_size=b))
    atlas = bt.TileAtlas.new(cfg, device)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas, str(tmp_path)).preprocess_tile(
        bt.PreprocessDataset(path="s", lod_range=range(0, lods)), bt.AssetServer().insert("s", src), atlas, defer_upload=True)
    st = pre.run_streamed_sharded(atlas, str(tmp_path), keep_queue=True)
    assert st["streamed"] and st["early_tiles"] == 256
    directory = atlas.attachment_directory(str(tmp_path), 0)
    for c, i in oracle.tiles():
        tile = np.fromfile(os.path.join(directory, f"{c[0]}_{c[1]}_{c[2]}_{c[3]}.bin"), dtype=np.uint16).reshape(T, T)
        assert np.array_equal(tile, oracle.tile(0, i)), c
    _ffi.check(L.bt_preprocessor_set_shard(pre._h, 1, 2))
    with pytest.raises(_ffi.BtError) as e:
        pre.run_streamed(atlas, str(tmp_path), keep_queue=True)
    assert e.value.status == -5 and "run_streamed_sharded" in str(e.value)
    with pytest.raises(_ffi.BtError) as e:  # both halves in one call need the communicator
        pre.run_streamed_sharded(atlas, str(tmp_path), keep_queue=True)
    assert e.value.status == -1
