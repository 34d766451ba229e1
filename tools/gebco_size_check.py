"""tools/gebco_size_check.py (about 30 GB of host memory, 20 GB on the device, ~35 s): a GEBCO_2023-sized planar source (86400 x 43200 R16 = 7.46 GB: byte offsets beyond 2^32), lod_count 8 -> 21845 tiles (11.4 GB atlas),
against the oracle: the index contract for every tile, a sample of tiles byte for byte (every 37th + the top of the pyramid)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _oracle as O
import bevy_terrain_amd as bt
W, H, lods = 86400, 43200, 8
device = bt.Device(0)
t0 = time.time()
ptr = device.synth_fbm_r16(W, H, 2023)
src = device.download(ptr, (H, W), np.uint16)
src[20000:20700, 60000:60900] = 0           # a no-data patch beyond the 4 GiB byte offset (row 20000 * 172800 B = 3.46 GB .. 3.58; and one lower)
src[30000:30300, 1000:1500] = 0             # 30000 * 172800 = 5.18 GB
device.free(ptr)
ptr = device.upload(src)
print("source ready", round(time.time() - t0, 1), "s", flush=True)
n_tiles = sum(4 ** l for l in range(lods))
cfg = bt.TerrainConfig(lod_count=lods, atlas_size=21900, path="terrains/gebco", model=bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 1.0))
cfg.add_attachment(bt.AttachmentConfig(name="height", texture_size=512, border_size=2, format=bt.AttachmentFormat.R16))
atlas = bt.TileAtlas.new(cfg, device)
pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
    bt.PreprocessDataset(attachment_index=0, path="g", lod_range=range(0, lods)), bt.AssetServer().insert("g", (ptr, W, H)), atlas)
t1 = time.time()
pre.run(atlas, keep_queue=True, profile=True)
print("run", round((time.time() - t1) * 1e3, 2), "ms host wall;", pre.stats(), [(l["kind"], round(l["avg_ms"], 3)) for l in pre.profile()], flush=True)
oracle = O.OracleAtlas(lods, 21900, False, [(512, 2, 1, O.FORMAT_R16)])
t2 = time.time()
oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(O.usable_cores())
print("oracle", round(time.time() - t2, 1), "s on", O.usable_cores(), "threads", flush=True)
assert [((c.side, c.lod, c.x, c.y), i) for c, i in atlas.tiles()] == oracle.tiles()
picked = sorted(set(range(0, n_tiles, 37)) | set(range(n_tiles - 341, n_tiles)))
# + every finest tile that touches the no-data patches
touch = [i for (c, i) in oracle.tiles() if c[1] == lods - 1 and ((60000 / W * 128 - 1 <= c[2] <= 60900 / W * 128 + 1 and 20000 / H * 128 - 1 <= c[3] <= 20700 / H * 128 + 1) or (c[2] <= 3 and 30000 / H * 128 - 1 <= c[3] <= 30300 / H * 128 + 1))]
picked = sorted(set(picked) | set(touch))
bad = 0
for i in picked:
    if not np.array_equal(atlas.download_tiles(0, i, 1)[0], oracle.tile(0, i)):
        bad += 1
        print("MISMATCH tile", i, oracle.tiles()[i], flush=True)
print("compared", len(picked), "tiles (", len(touch), "touching no-data ), mismatches", bad, flush=True)
# the kept queue once more (previous values fetched) and timing
pre.run(atlas, keep_queue=True)
bad2 = sum(0 if np.array_equal(atlas.download_tiles(0, i, 1)[0], oracle.tile(0, i)) else 1 for i in picked[::5])
print("re-run mismatches", bad2, flush=True)
import torch
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(3): pre.run(atlas, keep_queue=True, sync=False)
device.synchronize()
s.record(device.torch_stream)
for _ in range(10): pre.run(atlas, keep_queue=True, sync=False)
e.record(device.torch_stream); device.synchronize()
ms = s.elapsed_time(e) / 10
print("steady state", round(ms, 3), "ms/job =", round(n_tiles / ms / 1e3, 3), "M tiles/s; algorithmic", pre.stats()["algorithmic_bytes"], "B ->", round(pre.stats()["algorithmic_bytes"] / ms / 1e9, 3), "TB/s", flush=True)
sys.exit(1 if bad or bad2 else 0)
