"""tools/ratio_fuzz.py: random source-to-tile ratios in [0.55, 2.7] (x and y drawn separately) at T = 512 / 256, both formats, planar and cube, with and without no-data,
against the oracle: fresh, re-run, streamed.  usage: tools/ratio_fuzz.py <cases> <seed>"""
import os, sys, tempfile, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import _cases as K
import _oracle as O
import bevy_terrain_amd as bt
N, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
device = bt.Device(0)
bad = 0
for it in range(N):
    T = int(rng.choice([512, 512, 512, 256]))
    b = int(rng.choice([2, 2, 4]))
    fmt = O.FORMAT_R16 if rng.random() < 0.6 else O.FORMAT_RGBA8
    cube = bool(rng.random() < 0.2)
    lods = 2 if cube else int(rng.integers(2, 4))
    c = T - 2 * b
    extent = c << (lods - 1)
    rx, ry = rng.uniform(0.55, 2.7, 2)
    if rng.random() < 0.5: ry = rx
    W, H = max(16, int(extent * rx)), max(16, int(extent * ry))
    if cube: W = H = min(W, 1400)
    holes = float(rng.choice([0.0, 0.02, 0.2]))
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=64, path="terrains/rf", **({} if cube else dict(model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=K.FMT[fmt]))
    oracle = O.OracleAtlas(lods, 64, cube, [(T, b, 1, fmt)])
    server = bt.AssetServer()
    if cube:
        faces = [K.random_raster(fmt, H, W, seed * 1000 + it * 7 + s, holes=holes) for s in range(6)]
        paths = [f"f{s}" for s in range(6)]
        for path, f in zip(paths, faces): server.insert(path, f)
        oracle.clear_attachment(0).preprocess_spherical(0, faces, (0, lods)).run(16)
    else:
        src = K.random_raster(fmt, H, W, seed * 1000 + it, holes=holes)
        server.insert("src", src)
        oracle.clear_attachment(0).preprocess_tile(0, src, (0, lods)).run(16)
    def queue(atlas, root=None, defer=False):
        pre = bt.Preprocessor.new().clear_attachment(0, atlas, root)
        if cube: return pre.preprocess_spherical(bt.SphericalDataset(attachment_index=0, paths=paths, lod_range=range(0, lods)), server, atlas, defer_upload=defer)
        return pre.preprocess_tile(bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lods)), server, atlas, defer_upload=defer)
    tag = dict(it=it, T=T, b=b, fmt=fmt, cube=cube, lods=lods, W=W, H=H, rx=round(float(rx), 3), ry=round(float(ry), 3), holes=holes)
    try:
        atlas = bt.TileAtlas.new(cfg, device)
        pre = queue(atlas)
        pre.run(atlas, keep_queue=True)
        n = K.assert_atlas_equal(atlas, oracle)
        pre.run(atlas)
        assert K.assert_atlas_equal(atlas, oracle) == n
        root = tempfile.mkdtemp(prefix="bt_rf_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
        atlas2 = bt.TileAtlas.new(cfg, device)
        queue(atlas2, root, defer=True).run_streamed(atlas2, root)
        assert K.assert_atlas_equal(atlas2, oracle) == n
        shutil.rmtree(root, ignore_errors=True)
    except AssertionError as e:
        bad += 1
        print("MISMATCH", tag, str(e)[:200], flush=True)
    if it % 50 == 49: print(it + 1, "cases,", bad, "mismatches", flush=True)
print("done:", N, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
