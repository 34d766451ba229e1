"""oracle/_ref: the reference's OWN WGSL executed on the CPU (oracle/wgsl_ref) vs the hand-written oracle.

This is what pins parity: split.wgsl / downsample.wgsl / stitch.wgsl / preprocessing.wgsl / refine_tiles.wgsl /
prepare_prepass.wgsl / functions.wgsl are read UNMODIFIED from /root/reference, translated mechanically to C++
(wgsl2cpp.py) and run; the oracle (oracle/bt_oracle.c) is the same author's reading of the same files.  Equality here
means the reading is right; a difference is a finding (two are documented below: the T % 8 dispatch quirk and the
abstract-constant rule).

Everything here runs on the CPU.  In this container the library is rebuilt from /root/reference when stale; on the GPU
box the built oracle/_ref/libbt_wgslref.so travels with the snapshot (the tests skip if it is absent)."""
import ctypes as C
import math
import os
import subprocess
import sys

import numpy as np
import pytest

import _oracle as O
import _wgslref as W

pytestmark = pytest.mark.skipif(not W.available(), reason="oracle/_ref/libbt_wgslref.so not built (needs /root/reference)")

HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------------------------- the translator itself
def selftest(twice=False):
    f, u = np.zeros(32, np.float32), np.zeros(32, np.uint32)
    W.lib().wref_selftest.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    W.lib().wref_selftest(int(twice), f.ctypes.data, u.ctypes.data)
    return f, u


def test_translator_known_answers():
    """oracle/wgsl_ref/selftest/*.wgsl (ours): every value below is derived by hand / numpy.float32 from the WGSL rules —
    abstract-float constants fold in f64, `let` concretises, f32 ops round once, integer division by zero, masked
    shifts, saturating f32 -> u32, swizzles, flattening constructors, matrix * vector, override / #ifdef / pointers /
    shadowing / switch / loops / arrays"""
    f32 = np.float32
    f, u = selftest()
    exp_f = [
        f32(0.1 + 0.2), f32(0.1) + f32(0.2), f32(1.0) + f32(0.1 + 0.2), 1.5, 1.5, -1.5, 2.5, 2.0, 0.125,
        f32(3.0) / f32(5.0) + f32(0.0) / f32(5.0) + f32(4.0) / f32(5.0), 5.0, 234.0, 2.0, 741.0, 4.0, 1.0, f32(32.0) / f32(3.0), 2.5,
        33.0, np.sqrt(f32(2.0)),
    ]
    assert [float(x) for x in f[:20]] == [float(f32(x)) for x in exp_f]
    exp_u = [17, 0, 2, 0xFFFFFFFB, 45, 0xFFFFFFFD, 32768 | (65535 << 16), (255 << 8) | (127 << 16), 80, 60911, 44, 1, 21, 5, 4, 3, 7, 1]
    assert u[:18].tolist() == exp_u
    f2, _ = selftest(twice=True)
    assert f2[15] == 2.0  # the #ifdef TWICE branch


def test_library_reports_the_shader_files_it_executes():
    names = [s.split()[-1] for s in W.sources()]
    for needed in ("src/shaders/preprocess/split.wgsl", "src/shaders/preprocess/downsample.wgsl", "src/shaders/preprocess/stitch.wgsl",
                   "src/shaders/preprocess/preprocessing.wgsl", "src/shaders/tiling_prepass/refine_tiles.wgsl",
                   "src/shaders/tiling_prepass/prepare_prepass.wgsl", "src/shaders/functions.wgsl", "src/shaders/types.wgsl"):
        assert needed in names
    if os.path.isdir(W.REFERENCE):  # the digests are those of the files as they lie in the reference
        import hashlib
        for line in W.sources():
            digest, path = line.split()
            if path.startswith("src/"):
                assert hashlib.sha256(open(os.path.join(W.REFERENCE, path), "rb").read()).hexdigest() == digest


@pytest.mark.skipif(not os.path.isdir(W.REFERENCE), reason="needs /root/reference")
def test_generated_code_is_not_tracked():
    """nothing derived from the reference's sources enters the history: oracle/_ref/ is git-ignored"""
    root = os.path.dirname(HERE)
    if subprocess.run(["git", "rev-parse", "--is-inside-work-tree"], cwd=root, capture_output=True).returncode != 0:
        pytest.skip("not a git work tree (an exported copy of the repository)")
    tracked = subprocess.run(["git", "ls-files", "oracle/_ref"], cwd=root, capture_output=True, text=True).stdout.strip()
    assert tracked == ""
    assert subprocess.run(["git", "check-ignore", "-q", "oracle/_ref/gen_split.inc"], cwd=root).returncode == 0


# ---------------------------------------------------------------------------------------------- preprocessing
def raster(fmt, h, w, seed, holes=0.0):
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


def both(build):
    """run the same queue with the oracle's kernels and with the executed WGSL; return the two atlases"""
    out = []
    for ref in (False, True):
        a = build()
        if ref:
            W.attach(a)
        a.run(4)
        out.append(a)
    return out


def assert_same(a, r):
    assert a.tiles() == r.tiles()
    assert len(a.tiles()) > 0
    for c, i in a.tiles():
        x, y = a.tile(0, i), r.tile(0, i)
        if not np.array_equal(x, y):
            d = np.argwhere(x != y)
            raise AssertionError(f"tile {c}: {len(d)} texels differ, first {tuple(d[0])}: oracle {x[tuple(d[0])]} wgsl {y[tuple(d[0])]}")
    return len(a.tiles())


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
@pytest.mark.parametrize("T,b,lods,size,holes", [(16, 2, 3, 100, 0.0), (16, 2, 3, 100, 0.2), (32, 4, 4, 300, 0.05), (64, 2, 3, 257, 0.1),
                                                  (24, 3, 3, 90, 0.3), (8, 1, 4, 40, 0.1), (16, 1, 2, 7, 0.0), (40, 8, 2, 33, 0.5)])
def test_planar_jobs(fmt, T, b, lods, size, holes):
    """split + downsample + stitch over whole pyramids: magnifying and minifying resampling ratios, holes, wide borders"""
    src = raster(fmt, size, size + 3, 100 + T + lods, holes)

    def build():
        a = O.OracleAtlas(lods, 512, False, [(T, b, 1, fmt)])
        return a.clear_attachment(0).preprocess_tile(0, src, (0, lods))

    assert_same(*both(build))


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_subrect_datasets_overlays_and_lod_ranges(fmt):
    """dataset rectangles (inverse_mix, split.wgsl:30), `inside` (always true for centre pixels), keep-previous where the
    new dataset has no data (split.wgsl:37-42), partial lod ranges, datasets side by side"""
    T, b, lods = 16, 2, 4
    s0, s1, s2 = raster(fmt, 80, 96, 1, 0.1), raster(fmt, 50, 40, 2, 0.3), raster(fmt, 33, 70, 3, 0.0)

    def build():
        a = O.OracleAtlas(lods, 512, False, [(T, b, 1, fmt)])
        a.clear_attachment(0)
        a.preprocess_tile(0, s0, (0, lods), top_left=(0.0, 0.0), bottom_right=(0.5, 1.0))
        a.preprocess_tile(0, s2, (0, lods), top_left=(0.5, 0.0), bottom_right=(1.0, 1.0))
        a.preprocess_tile(0, s1, (1, lods), top_left=(0.2, 0.3), bottom_right=(0.7, 0.9))
        return a

    assert_same(*both(build))


def test_random_rects_both_formats():
    rng = np.random.default_rng(77)
    for k in range(12):
        fmt = O.FORMAT_R16 if k % 2 else O.FORMAT_RGBA8
        T = int(rng.choice([8, 16, 24, 32]))
        b = int(rng.integers(1, T // 4 + 1))
        lods = int(rng.integers(1, 4))
        x0, y0 = rng.random(2) * 0.5
        x1, y1 = x0 + 0.1 + rng.random() * 0.4, y0 + 0.1 + rng.random() * 0.4
        src = raster(fmt, int(rng.integers(5, 120)), int(rng.integers(5, 120)), 500 + k, float(rng.choice([0.0, 0.1, 0.4])))

        def build():
            a = O.OracleAtlas(lods, 512, False, [(T, b, 1, fmt)])
            return a.clear_attachment(0).preprocess_tile(0, src, (0, lods), top_left=(float(x0), float(y0)), bottom_right=(float(x1), float(y1)))

        assert_same(*both(build))


@pytest.mark.parametrize("fmt", [O.FORMAT_R16, O.FORMAT_RGBA8])
def test_cube_jobs_all_edge_orientations(fmt):
    """six faces with independent random data: every one of the 24 face-edge adjacencies (EVEN / ODD tables of
    project_to_side, stitch.wgsl:12-51) moves texels that differ from every other candidate, at two LODs"""
    T, b, lods, w = 16, 2, 3, 48
    faces = [raster(fmt, w, w, 900 + s, 0.05) for s in range(6)]

    def build():
        a = O.OracleAtlas(lods, 512, True, [(T, b, 1, fmt)])
        return a.clear_attachment(0).preprocess_spherical(0, faces, (0, lods))

    a, r = both(build)
    assert assert_same(a, r) == 6 * 21
    # the seams were really exercised: aprons of edge tiles hold other faces' texels (not the clamp-to-own fallback)
    edge = next(i for c, i in a.tiles() if c[1] == 2 and c[2] == 0)
    t = a.tile(0, edge)
    assert not np.array_equal(t[b:T - b, 0], t[b:T - b, b])


def test_reference_dispatch_covers_only_multiples_of_eight_rows():
    """FINDING (documented): the reference dispatches (entries_per_side / 8, texture_size / 8) workgroups of
    8 x 8 (gpu_tile_atlas.rs:101), so for a texture_size that is not a multiple of 8 its shaders never write the last
    texture_size % 8 rows of a tile.  The oracle and the product process every row; they agree with the reference on all
    rows the reference writes.  Every texture size the reference's examples use is a multiple of 8."""
    T, b, lods = 20, 2, 2
    src = raster(O.FORMAT_R16, 100, 100, 5)

    def build():
        a = O.OracleAtlas(lods, 64, False, [(T, b, 1, O.FORMAT_R16)])
        return a.clear_attachment(0).preprocess_tile(0, src, (1, lods))  # the finest LOD only: split + stitch, no pyramid

    a, r = both(build)
    covered = (T // 8) * 8
    for c, i in a.tiles():
        x, y = a.tile(0, i), r.tile(0, i)
        rows = sorted(set(np.argwhere(x != y)[:, 0].tolist()))
        # the rows the reference never dispatches differ, and so do the top aprons that stitch copies from a neighbour's
        # never-written bottom rows; every centre row the reference does write is identical
        assert rows and all(r >= covered or r < b for r in rows), (c, rows)
        assert np.array_equal(x[b:covered], y[b:covered])


# ---------------------------------------------------------------------------------------------- tiling prepass
def spiral(n, radius, h0, h1, seed=99):
    rng = np.random.default_rng(seed)
    for i in range(n):
        t = i / max(n - 1, 1)
        a = 2 * math.pi * 3 * t + rng.random() * 0.01
        r = radius * (1 - 0.9 * t)
        yield (r * math.cos(a), h0 + (h1 - h0) * t, r * math.sin(a))


MODELS = {
    "planar": dict(kind="planar", position=(0, 0, 0), a=1000.0, min_height=0.0, max_height=250.0),
    "sphere": dict(kind="spherical", position=(0, 0, 0), a=6371000.0, min_height=-12000.0, max_height=9000.0),
    "ellipsoid": dict(kind="ellipsoidal", position=(10.0, -20.0, 30.0), a=6378137.0, b=6356752.314245, min_height=-12000.0, max_height=9000.0),
}


def camera_positions(name, n, seed):
    if name == "planar":
        return list(spiral(n, 700.0, 900.0, 130.0, seed))
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        out.append(tuple(d * (6371000.0 + 10 ** rng.uniform(2.0, 6.6))))
    return out


@pytest.mark.parametrize("name", list(MODELS))
def test_refine_lists_and_distances(name):
    """prepare_root, 30 x (refine_tiles, prepare_next), refine_tiles, prepare_render executed from the WGSL vs orc_refine:
    identical final LIST (order of a sequential run), indirect args, per-pass tile counts; and the f32 view distance of
    should_be_divided (functions.wgsl:73-131) bit for bit on every final tile"""
    model = O.make_model(**MODELS[name])
    vc = O.make_view_config(geometry_tile_count=300000)
    tiles_total = distances = 0
    for k, pos in enumerate(camera_positions(name, 14, 1234)):
        v = O.view_state_from_config(model, vc, pos, 100.0 + 150.0 * k)
        a, ia, pa = O.refine(v)
        b, ib, pb = W.refine(v)
        assert np.array_equal(a, np.array(b, dtype=np.uint32).reshape(-1, 4)), (name, k)
        assert ia == ib and pa == pb
        tiles_total += len(a)
        for t in a[:: max(1, len(a) // 300)]:
            r1, d1 = O.should_be_divided(v, tuple(int(x) for x in t))
            r2, d2 = W.should_be_divided(v, tuple(int(x) for x in t))
            assert r1 == r2 and np.float32(d1).tobytes() == np.float32(d2).tobytes(), (name, k, t, d1, d2)
            distances += 1
    assert tiles_total > 2000 and distances > 1000


def test_refine_small_buffers_and_refinement_counts():
    """tiles still dividing on the last pass are dropped; tiny refinement counts; the ping-pong buffer at its limit"""
    model = O.make_model(**MODELS["sphere"])
    for refinement_count, tile_count in ((0, 64), (1, 64), (3, 4096), (7, 20000)):
        vc = O.make_view_config(geometry_tile_count=tile_count, refinement_count=refinement_count)
        v = O.view_state_from_config(model, vc, (0.0, 6371000.0 + 5000.0, 1000.0), 0.0)
        try:
            a = O.refine(v)
        except OverflowError:
            with pytest.raises(OverflowError):
                W.refine(v)
            continue
        b = W.refine(v)
        assert np.array_equal(a[0], np.array(b[0], dtype=np.uint32).reshape(-1, 4)) and a[1:] == b[1:]


@pytest.mark.parametrize("name", list(MODELS))
def test_final_set_is_a_per_tile_predicate_of_the_executed_divide_test(name):
    """What bt_tiling_prepass_run_unordered relies on (bt_refine.hip), stated against the reference's own WGSL: the
    final list of the executed refine_tiles / prepare_prepass schedule is, as a set, exactly the tiles that (a) are reached
    — every ancestor satisfies the executed should_be_divided — (b) do not divide themselves and (c) have lod <=
    refinement_count; the children of tiles that still divide in the last pass appear nowhere.  Depth-first from the roots,
    one executed divide test per visited tile; also with a refinement_count that cuts the tree short."""
    model = O.make_model(**MODELS[name])
    pos = camera_positions(name, 3, 77)[-1]
    for rc in (30, 3):
        v = O.view_state_from_config(model, O.make_view_config(geometry_tile_count=300000, refinement_count=rc), pos, 120.0)
        schedule, _, passes = W.refine(v)
        final, visited = [], 0
        stack = [(s, 0, 0, 0) for s in range(6 if v.spherical else 1)]
        while stack:
            side, lod, x, y = stack.pop()
            visited += 1
            if not W.should_be_divided(v, (side, lod, x, y))[0]:
                final.append((side, lod, x, y))
            elif lod < rc:
                stack.extend((side, lod + 1, 2 * x + (i & 1), 2 * y + (i >> 1)) for i in range(4))
        assert visited == sum(passes)
        assert len(final) == len(schedule) == len(set(schedule)) and sorted(final) == sorted(schedule), (name, rc)


@pytest.mark.skipif(not os.path.isdir(W.REFERENCE), reason="needs /root/reference to re-translate")
def test_abstract_constant_rule_is_observable_only_below_one_ulp(tmp_path):
    """FINDING: `const C_SQR = 0.87 * 0.87;` (functions.wgsl:12).  naga 0.20 (bevy 0.14.0's shader compiler,
    Cargo.toml:18) concretises an untyped const at its declaration, so `1.0 + C_SQR` is an f32 sum — what the oracle and
    the product compute.  Under the WGSL specification's rule (abstract consts stay abstract; later naga releases) the
    sum is rounded once from f64 and the warp denominator moves by 1 ULP.  Re-translate with that rule and measure: the
    distances move by at most a few ULP and the final tile lists do not change on the scripted cameras."""
    ref_dir, out = W.REF_DIR, str(tmp_path)
    sh = os.path.join(W.REFERENCE, "src", "shaders")
    env = dict(os.environ, WGSL2CPP_ABSTRACT_CONSTS="1")
    subprocess.check_call([sys.executable, os.path.join(ref_dir, "wgsl2cpp.py"), os.path.join(out, "gen.inc"), "PrepassSpherical", "SPHERICAL",
                           sh + "/tiling_prepass/refine_tiles.wgsl," + sh + "/tiling_prepass/prepare_prepass.wgsl", sh + "/tiling_prepass", sh,
                           os.path.join(ref_dir, "bevy_supplied")], env=env)
    spec = open(os.path.join(out, "gen.inc")).read()
    pinned = open(os.path.join(O.ORACLE_DIR, "_ref", "gen_prepass_spherical.inc")).read()
    assert "C_SQR__functions = (AF(0.87) * AF(0.87))" in spec and "C_SQR__functions = w_concretize((AF(0.87) * AF(0.87)))" in pinned
    f32 = np.float32
    assert f32(1.0) + f32(0.87 * 0.87) != f32(1.0 + 0.87 * 0.87)  # the 1-ULP difference itself
    assert f32(0.87 * 0.87) == f32(0.87) * f32(0.87)  # C_SQR alone is the same either way
