"""A compiled, non-Python host drives the path through the C ABI (tests/abi_job.c: the stand-in for the Rust `extern "C"` block of
INTEGRATION.md, which cannot be compiled here): preprocess_tile on a host raster -> run -> save, then the tiling prepass of one
view.  Everything it writes is compared with the oracle: the .bin tile files byte for byte, config.tc as a set of coordinates,
the final tile list in order, the indirect arguments."""
import json
import os
import struct
import subprocess

import numpy as np
import pytest

import _oracle as O
import bevy_terrain_amd as bt
from bevy_terrain_amd import _ffi

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def raster_of_abi_job():
    x, y = np.meshgrid(np.arange(300, dtype=np.uint64), np.arange(300, dtype=np.uint64))
    src = (1 + (x * 131 + y * 71 + (x * y) % 97) % 60000).astype(np.uint16)
    src[40:60, 100:130] = 0
    return src


def test_c_host_runs_preprocess_save_and_tiling_prepass(tmp_path):
    exe = str(tmp_path / "abi_job")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_job.c"), "-ldl", "-o", exe])
    out_dir = str(tmp_path / "assets")
    os.makedirs(out_dir)
    report = json.loads(subprocess.check_output([exe, _ffi.LIB_PATH, out_dir], text=True))
    assert report["tiles"] == 21 and report["atlas_tiles"] == 21 and report["fused_jobs"] == 1

    # --- the tile files and config.tc against the oracle's (queue order, atlas indices, every byte)
    src = raster_of_abi_job()
    oracle = O.OracleAtlas(3, 64, False, [(64, 2, 1, O.FORMAT_R16)])
    oracle.clear_attachment(0).preprocess_tile(0, src, (0, 3)).run(4)
    odir = str(tmp_path / "oracle")
    os.makedirs(odir)
    oracle.save_attachment(0, odir)
    d = os.path.join(out_dir, "terrains/abi_job/data/height")
    names = sorted(os.listdir(d))
    assert names == sorted(os.listdir(odir)) and len(names) == 21
    for n in names:
        assert open(os.path.join(d, n), "rb").read() == open(os.path.join(odir, n), "rb").read(), n
    tc = open(os.path.join(out_dir, "terrains/abi_job/config.tc"), "rb").read()
    assert set(O.tc_decode(tc)) == {c for c, _ in oracle.tiles()}

    # --- the tiling prepass: the list in the reference's order, and prepare_render's indirect arguments
    from test_gpu_refine import oracle_view

    raw = open(os.path.join(out_dir, "final_tiles.bin"), "rb").read()
    count, = struct.unpack_from("<I", raw, 0)
    indirect = list(struct.unpack_from("<4I", raw, 4))
    tiles = np.frombuffer(raw, dtype=np.uint32, offset=20).reshape(-1, 4)
    assert count == len(tiles) == report["final_tiles"] and count > 100
    model = bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 250.0)
    cfg = bt.TerrainViewConfig(geometry_tile_count=100000)
    view = bt.make_view_state(model, cfg, (120.0, 260.0, -75.0), approximate_height=0.0)
    exp, exp_indirect, _ = O.refine(oracle_view(view))
    assert np.array_equal(tiles, exp) and indirect == exp_indirect
