"""The boundary's threading rule: "thread-safe per context (one context may be driven from the render thread
while another thread polls)".  The reference runs its main world and its render world on different threads (pipelined
rendering: src/preprocess/mod.rs:272-293 extracts / prepares / runs the preprocess node in the render app, src/plugin.rs:78-108
the tiling prepass, while src/plugin.rs:46-56 updates tile trees and atlases in the main app).  Here: two HOST THREADS, two
contexts on one GPU — thread A loops bt_preprocessor_run on context 1, thread B loops bt_frame_update / bt_tiling_prepass_read
on context 2, a third thread polls bt_last_error and provokes errors of its own — and every output equals the single-threaded
run of the same work.  (ctypes releases the GIL around every call into the library: the calls really overlap.)"""
import threading

import numpy as np
import pytest

import _cases as K
import _oracle as O
import bevy_terrain_amd as bt
from test_gpu_tile_tree import MODELS, build_terrain, camera_path

pytestmark = pytest.mark.gpu


def _stream_instance(device, root, cfg, model, lods, T, b, vc):
    scfg = bt.TerrainConfig(lod_count=lods, atlas_size=256, path=cfg.path, model=model)
    scfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=T, border_size=b, format=bt.AttachmentFormat.R16, mip_level_count=3))
    atlas = bt.TileAtlas.new(scfg, device)
    atlas.load_tile_config(root)
    return atlas, bt.TileTree.new(atlas, vc), bt.TilingPrepass(device, vc.geometry_tile_count)


def _frames(root, atlas, tree, prepass, path, unordered):
    """the per-frame chain of plugin.rs:46-56 along a camera path; everything a renderer would consume, frame by frame"""
    out = []
    for pos in path:
        atlas.update(root)
        info = tree.frame_update(pos, prepass, unordered=unordered)
        entries, origins, coords, flags = tree.read()
        tiles, indirect = prepass.read()
        tiles = sorted(map(tuple, tiles)) if unordered else [tuple(t) for t in tiles]
        out.append((info.released_count, info.requested_count, info.apply_status, entries.tobytes(), origins.tobytes(), coords.tobytes(), flags.tobytes(),
                    tiles, tuple(indirect)))
    return out


@pytest.mark.parametrize("unordered", [False, True])
def test_two_host_threads_two_contexts_and_a_polling_thread(tmp_path, unordered):
    index = 0
    dev_a, dev_b, dev_ref = bt.Device(index), bt.Device(index), bt.Device(index)  # three contexts = three HIP streams on one GPU
    # context 1's work: a preprocess job with no-data (fused plan), run over and over
    T, b, lods, fmt = 256, 2, 4, O.FORMAT_R16
    src = K.random_raster(fmt, 1000, 1100, 4242, holes=0.02)
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=128, path="terrains/threads", model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
    atlas1 = bt.TileAtlas.new(cfg, dev_a)
    server = bt.AssetServer().insert("src", src)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas1).preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods)), server, atlas1)
    # context 2's work: a streaming terrain, its tile tree and its tiling prepass along a camera path
    model, _ = MODELS["planar"]
    slods, sT, sb = 4, 32, 2
    root, scfg, _ = build_terrain(dev_ref, tmp_path, model, slods, sT, sb)
    vc = bt.TerrainViewConfig(tree_size=4, load_distance=1.2, blend_distance=1.0, geometry_tile_count=20000)
    path = camera_path("planar", 40, seed=11)
    expected = _frames(root, *_stream_instance(dev_ref, root, scfg, model, slods, sT, sb, vc), path, unordered)  # single-threaded
    atlas2, tree2, prepass2 = _stream_instance(dev_b, root, scfg, model, slods, sT, sb, vc)

    errors, got, polled = [], [], [0, 0]
    text_before = bt._ffi.lib().bt_last_error()
    start, done = threading.Barrier(3), threading.Event()

    def thread_a():
        try:
            start.wait()
            for i in range(120):
                pre.run(atlas1, keep_queue=True, sync=(i % 16 == 15))
            dev_a.synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append(("A", repr(e)))

    def thread_b():
        try:
            start.wait()
            got.extend(_frames(root, atlas2, tree2, prepass2, path, unordered))
        except Exception as e:  # noqa: BLE001
            errors.append(("B", repr(e)))

    def thread_c():
        # bt_last_error is per thread: this thread's own failures (an atlas whose border leaves no centre, a null handle) are
        # the only text it ever sees, and they never show up in the status of the calls the other two threads make
        try:
            lib = bt._ffi.lib()
            start.wait()
            while not done.is_set():
                polled[0] += 1
                assert lib.bt_preprocessor_run(None, None, 0) != 0  # BT_ERR_INVALID_ARGUMENT, no text required
                bad = bt.TerrainConfig(lod_count=3, atlas_size=8, path="terrains/bad", model=model)
                bad.add_attachment(bt.AttachmentConfig(name="x", texture_size=6, border_size=4, format=bt.AttachmentFormat.R16))  # no centre left
                try:
                    bt.TileAtlas.new(bad, dev_ref)
                except bt._ffi.BtError as e:
                    polled[1] += 1
                    text = lib.bt_last_error().decode(errors="replace")
                    assert text and text in str(e)
                else:
                    errors.append(("C", "an atlas whose border leaves no centre was accepted"))
        except Exception as e:  # noqa: BLE001
            errors.append(("C", repr(e)))

    threads = [threading.Thread(target=f) for f in (thread_a, thread_b, thread_c)]
    for t in threads:
        t.start()
    threads[0].join()
    threads[1].join()
    done.set()
    threads[2].join()
    assert not errors, errors
    assert polled[0] > 0 and polled[1] > 0
    # the calling thread made no failing call: its error text is untouched by thread C's failures
    assert bt._ffi.lib().bt_last_error() == text_before
    assert len(got) == len(expected)
    for frame, (g, e) in enumerate(zip(got, expected)):
        assert g == e, frame
    assert K.assert_atlas_equal(atlas1, K.oracle_planar(src, lods, T, b, fmt)) == 85
