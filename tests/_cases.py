"""Shared builders: run the same preprocess job through the product (HIP, via the mirrored reference
API) and through the oracle, and compare every tile byte for byte."""
import numpy as np

import _oracle as O
import bevy_terrain_amd as bt

FMT = {O.FORMAT_R16: bt.AttachmentFormat.R16, O.FORMAT_RGBA8: bt.AttachmentFormat.Rgba8}


def random_raster(fmt, h, w, seed, holes=0.0):
    rng = np.random.default_rng(seed)
    if fmt == O.FORMAT_R16:
        src = rng.integers(1, 65536, size=(h, w), dtype=np.uint16)
        if holes:
            src[rng.random(src.shape) < holes] = 0
    else:
        src = rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
        src[..., 0] = np.maximum(src[..., 0], 1)
        if holes:
            src[..., 0][rng.random(src.shape[:2]) < holes] = 0
    return src


def smooth_raster(h, w, seed, device=None):
    """fBm heightmap in [1, 65535]; large sizes come from the HIP generator (which the GPU tests check
    against the integer numpy model), small ones from the model itself."""
    if device is not None:
        ptr = device.synth_fbm_r16(w, h, seed)
        out = device.download(ptr, (h, w), np.uint16)
        device.free(ptr)
        return out
    import _model as M
    return M.fbm_u16(w, h, seed)


def product_planar(device, src, lod_count, T, b, fmt, atlas_size=128, generic=False, mips=1, reference_dispatch=False, **ds):
    cfg = bt.TerrainConfig(lod_count=lod_count, atlas_size=atlas_size, path="terrains/test",
                           model=bt.TerrainModel.planar((0, 0, 0), 1000.0, 0.0, 1.0))
    cfg.add_attachment(bt.AttachmentConfig(name="att", texture_size=T, border_size=b, format=FMT[fmt],
                                           mip_level_count=mips))
    atlas = bt.TileAtlas.new(cfg, device)
    server = bt.AssetServer().insert("src", src)
    pre = bt.Preprocessor.new().clear_attachment(0, atlas).preprocess_tile(
        bt.PreprocessDataset(attachment_index=0, path="src", lod_range=range(0, lod_count), **ds), server, atlas)
    pre.run(atlas, generic=generic, reference_dispatch=reference_dispatch)
    return atlas, pre


def oracle_planar(src, lod_count, T, b, fmt, atlas_size=128, threads=8, **ds):
    a = O.OracleAtlas(lod_count, atlas_size, False, [(T, b, 1, fmt)])
    a.clear_attachment(0).preprocess_tile(0, src, (0, lod_count), **ds).run(threads)
    return a


def reference_kernels(oracle_atlas):
    """Checker for the BASELINE configs and smoke(): the oracle's queue driver with every Split / Downsample / Stitch task
    executed by the REFERENCE'S OWN WGSL (oracle/_ref, see tests/_wgslref.py) instead of the hand-written kernels.  The
    built library travels to the GPU box; its absence is an error, not a silent fallback."""
    import _wgslref as W

    assert W.available(), "oracle/_ref/libbt_wgslref.so missing: run `make -C oracle/wgsl_ref` where /root/reference exists"
    return W.attach(oracle_atlas)


def assert_atlas_equal(atlas, oracle, attachment=0):
    ours = atlas.tiles()
    theirs = oracle.tiles()
    # the integer tile-index contract: same coordinates at the same atlas indices
    assert [((c.side, c.lod, c.x, c.y), i) for c, i in ours] == theirs
    n = len(ours)
    if n == 0:
        return 0
    used = max(i for _, i in ours) + 1
    guard = min(2, atlas.config.atlas_size - used)  # the layers behind the last tile: nothing may have written them (zero since bt_atlas_create)
    data = atlas.download_tiles(attachment, 0, used + guard)
    assigned = {i for _, i in ours}
    stray = [i for i in range(used + guard) if i not in assigned and data[i].any()]
    assert not stray, f"layers {stray[:5]} hold no tile and are not zero"
    bad = []
    for coord, idx in theirs:
        exp = oracle.tile(attachment, idx)
        if not np.array_equal(data[idx], exp):
            diff = np.argwhere(data[idx] != exp)
            bad.append((coord, len(diff), tuple(diff[0]), data[idx][tuple(diff[0])], exp[tuple(diff[0])]))
    assert not bad, f"{len(bad)}/{n} tiles differ, first: {bad[:3]}"
    return n
