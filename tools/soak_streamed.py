"""Soak of the streamed pipeline: random small jobs, streamed (deferred rasters, bands, saver thread, three queues) against serial (run + save)
into two directories, compared file by file.  usage: tools/soak_streamed.py <iterations> <seed>"""
import filecmp, os, shutil, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bevy_terrain_amd as bt

N, seed = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
device = bt.Device(0)
parent = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
t0 = time.time()
bad = 0
streamed_runs = bands_total = 0
for it in range(N):
    T = int(rng.choice([64, 128, 256, 512]))
    lods = int(rng.integers(2, 5))
    c = T - 4
    extent = c << (lods - 1)
    natt = int(rng.integers(1, 3))
    cube = bool(rng.random() < 0.15 and T <= 128)
    fmts = [bt.AttachmentFormat.R16 if rng.random() < 0.6 else bt.AttachmentFormat.Rgba8 for _ in range(natt)]
    cfg = bt.TerrainConfig(lod_count=lods, atlas_size=2100 if cube else 400, path="terrains/soak", model=bt.TerrainModel.sphere((0, 0, 0), 1.0, 0.0, 1.0) if cube else bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    for i, f in enumerate(fmts):
        cfg.add_attachment(bt.AttachmentConfig(name=f"a{i}", texture_size=T, border_size=2, format=f))
    def raster(f, n):
        if f == bt.AttachmentFormat.R16:
            a = rng.integers(0 if rng.random() < 0.3 else 1, 65536, size=(n, n), dtype=np.uint16)
            if rng.random() < 0.3: a[rng.random((n, n)) < 0.05] = 0
            return a
        return rng.integers(0, 256, size=(n, n, 4), dtype=np.uint8)
    n = int(np.clip(extent * rng.uniform(0.5, 1.6), 16, 2600))
    server = bt.AssetServer()
    for i, f in enumerate(fmts):
        if cube:
            for s in range(6): server.insert(f"s{i}_{s}", raster(f, min(n, 600)))
        else:
            server.insert(f"s{i}", raster(f, n))
    roots = []
    for streamed in (False, True):
        root = tempfile.mkdtemp(prefix="bt_soak_", dir=parent)
        roots.append(root)
        atlas = bt.TileAtlas.new(cfg, device)
        pre = bt.Preprocessor.new()
        for i in range(natt): pre.clear_attachment(i, atlas, root)
        for i in range(natt):
            if cube:
                pre.preprocess_spherical(bt.SphericalDataset(attachment_index=i, paths=[f"s{i}_{s}" for s in range(6)], lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
            else:
                pre.preprocess_tile(bt.PreprocessDataset(attachment_index=i, path=f"s{i}", lod_range=range(0, lods)), server, atlas, defer_upload=streamed)
        if streamed:
            st = pre.run_streamed(atlas, root)
            streamed_runs += int(st['streamed']); bands_total += st['bands']
        else:
            pre.run(atlas)
            pre.save(atlas, root)
        pre.close(); atlas.close()
    cmp = filecmp.dircmp(roots[0], roots[1])
    def differs(d):
        if d.left_only or d.right_only or d.funny_files: return True
        match, mismatch, errors = filecmp.cmpfiles(d.left, d.right, d.common_files, shallow=False)
        if mismatch or errors: return True
        return any(differs(s) for s in d.subdirs.values())
    if differs(cmp):
        bad += 1
        print("MISMATCH iteration", it, dict(T=T, lods=lods, natt=natt, cube=cube, n=n, fmts=[f.value for f in fmts]), flush=True)
    for r in roots: shutil.rmtree(r, ignore_errors=True)
    if it % 100 == 99: print(f"{it + 1} iterations, {bad} mismatches, {streamed_runs} streamed with {bands_total} bands, {time.time() - t0:.0f} s", flush=True)
print(f"done: {N} iterations, {bad} mismatches, {streamed_runs} streamed with {bands_total} bands, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
