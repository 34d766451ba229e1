"""CPU-side checks of the product library: ABI surface, TileCoordinate maths, TC codec, view maths.
No kernel is launched here (there is no GPU in the build container)."""
import math
import os
import re
import subprocess

import numpy as np
import pytest

import _oracle as O
import bevy_terrain_amd as bt
from bevy_terrain_amd import _ffi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    declared = _ffi.header_symbols()
    assert len(declared) >= 45
    out = subprocess.check_output(["nm", "-D", "--defined-only", _ffi.LIB_PATH], text=True)
    exported = set(re.findall(r" T (bt_[a-z0-9_]+)", out))
    assert declared <= exported, sorted(declared - exported)
    assert declared == set(_ffi.PROTOTYPES), sorted(declared ^ set(_ffi.PROTOTYPES))
    assert _ffi.lib().bt_abi_version() == _ffi.header_abi_version()


def test_graft_entry_build_runs_end_to_end():
    """build() is what the driver calls every round: it must succeed on a correct tree (round 3 shipped it asserting a stale
    ABI literal).  An up-to-date tree makes this a no-op `make`."""
    import __graft_entry__ as g
    g.build()


def test_documents_agree_with_the_header_on_abi_version_and_symbol_count():
    version, count = _ffi.header_abi_version(), len(_ffi.header_symbols())
    for name in ("DESIGN.md", "INTEGRATION.md"):
        text = open(os.path.join(ROOT, name)).read()
        versions = set(int(v) for v in re.findall(r"ABI (?:v|version )(\d+)", text))
        assert versions <= {version} and (versions or name != "DESIGN.md"), (name, versions, version)
        counts = set(int(c) for c in re.findall(r"(\d+) (?:symbols|entry points|in all)", text))
        assert counts and counts <= {count}, (name, counts, count)


def test_struct_layouts_match_reference_gpu_layouts():
    import ctypes as C
    assert C.sizeof(_ffi.TileCoordinateC) == 16  # types.wgsl:25-29
    assert C.sizeof(_ffi.AtlasTileC) == 32       # preprocessing.wgsl:16-22
    assert C.sizeof(_ffi.IndirectC) == 16        # terrain_view_bind_group.rs:65-71


def test_no_gpu_fails_loudly_not_silently():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_ffi.BtError):
        bt.Device(0)


@pytest.mark.parametrize("spherical", [False, True])
def test_tile_coordinate_matches_oracle(spherical):
    for lod in range(0, 4):
        n = 1 << lod
        for side in range(6 if spherical else 1):
            for x in range(n):
                for y in range(n):
                    c = bt.TileCoordinate(side, lod, x, y)
                    ours = [(t.side, t.lod, t.x, t.y) for t in c.neighbours(spherical)]
                    assert ours == O.neighbours((side, lod, x, y), spherical)
                    assert [(t.side, t.lod, t.x, t.y) for t in c.children()] == O.children((side, lod, x, y))
    c = bt.TileCoordinate(3, 5, 17, 9)
    assert str(c) == "3_5_17_9" and c.path("a/b", "bin") == "a/b/3_5_17_9.bin"
    assert c.parent() == bt.TileCoordinate(3, 4, 8, 4)


def test_tc_codec_matches_oracle_bytes_and_bincode_varints():
    coords = [(0, 0, 0, 0), (5, 4, 15, 250), (1, 9, 251, 511), (2, 17, 65535, 65536), (3, 30, 2 ** 30 - 1, 70000)]
    ours = bt.tc_encode([bt.TileCoordinate(*c) for c in coords])
    assert ours == O.tc_encode(coords)
    # hand-computed bincode-2 standard encoding: len=5, then varints (251 -> 0xFB + u16, 65536 -> 0xFC + u32)
    assert ours[:5] == bytes([5, 0, 0, 0, 0])
    assert bytes([0xFB, 0xFB, 0x00]) in ours and bytes([0xFC, 0x00, 0x00, 0x01, 0x00]) in ours
    assert [(t.side, t.lod, t.x, t.y) for t in bt.tc_decode(ours)] == coords
    assert bt.tc_decode(bt.tc_encode([])) == []
    with pytest.raises(ValueError):
        bt.tc_decode(b"\x02\x01")


def test_view_state_derivation():
    # planar example (examples/minimal.rs:6-7,32): side 1000, heights 0..250
    model = bt.TerrainModel.planar((0.0, 0.0, 0.0), 1000.0, 0.0, 250.0)
    v = bt.make_view_state(model, bt.TerrainViewConfig(), (100.0, 300.0, -200.0))
    assert v.spherical == 0 and v.refinement_count == 30 and v.vertices_per_tile == 2 * 16 * 18
    assert v.subdivision_distance == pytest.approx(16.0 * 500.0 * 1.1)
    assert v.approximate_height == 125.0
    u, w = (100.0 / 1000.0 + 0.5) * 1024, (-200.0 / 1000.0 + 0.5) * 1024
    assert (v.sides[0].view_xy[0], v.sides[0].view_xy[1]) == (int(u), int(w))
    assert v.sides[0].view_uv[0] == pytest.approx(u - int(u), abs=1e-6)
    # sphere: the view side gets its own uv, neighbours get a fixed edge coordinate
    sphere = bt.TerrainModel.sphere((0.0, 0.0, 0.0), 6371000.0, -12000.0, 9000.0)
    v = bt.make_view_state(sphere, bt.TerrainViewConfig(), (7e6, 1e5, -2e5))
    assert v.spherical == 1
    side3 = (v.sides[3].view_xy[0] / 1024.0, v.sides[3].view_xy[1] / 1024.0)
    assert 0.4 < side3[0] < 0.6 and 0.4 < side3[1] < 0.6  # +x face, near its centre
    # the opposite face (-x, side 0) takes both coordinates over; the four adjacent faces get one pinned to an edge
    for side in (1, 2, 4, 5):
        assert v.sides[side].view_xy[0] in (0, 1024) or v.sides[side].view_xy[1] in (0, 1024)


def test_c99_consumer_of_the_header_and_ctypes_mirror_layouts(tmp_path):
    """tests/abi_consumer.c includes the header as C99, static-asserts every struct layout, loads the library with
    dlopen and calls through it; its printed layouts must equal the hand-typed ctypes mirror field by field."""
    import ctypes as C
    import json

    exe = str(tmp_path / "abi_consumer")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Wextra", "-Werror", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "abi_consumer.c"), "-ldl", "-o", exe])
    out = json.loads(subprocess.check_output([exe, _ffi.LIB_PATH], text=True))
    assert out["abi_version"] == _ffi.header_abi_version() and out["done"]
    assert out["ctx_create"] in (0, -1, -3)  # 0 on a GPU box; a clean error status (with a message) without one
    assert out["view_state"] == [0, 614, 307, 2 * 16 * 18, 8]
    mirror = {"bt_tile_coordinate": _ffi.TileCoordinateC, "bt_atlas_tile": _ffi.AtlasTileC, "bt_attachment_config": _ffi.AttachmentConfigC,
              "bt_terrain_config": _ffi.TerrainConfigC, "bt_raster": _ffi.RasterC, "bt_preprocess_dataset": _ffi.PreprocessDatasetC,
              "bt_spherical_dataset": _ffi.SphericalDatasetC, "bt_tile_tree_entry": _ffi.TileTreeEntryC, "bt_run_stats": _ffi.RunStatsC,
              "bt_stream_stats": _ffi.StreamStatsC, "bt_shard_range": _ffi.ShardRangeC, "bt_launch_profile": _ffi.LaunchProfileC, "bt_side_parameter": _ffi.SideParameterC,
              "bt_view_state": _ffi.ViewStateC, "bt_indirect": _ffi.IndirectC, "bt_terrain_model": _ffi.TerrainModelC,
              "bt_terrain_view_config": _ffi.TerrainViewConfigC}
    checked = 0
    for name, cls in mirror.items():
        layout = out[name]
        assert layout.pop("__size__")[1] == C.sizeof(cls), name
        assert set(layout) == {f[0] for f in cls._fields_}, name
        for field, (offset, size) in layout.items():
            d = getattr(cls, field)
            assert (d.offset, d.size) == (offset, size), (name, field)
            checked += 1
    assert checked > 80


def test_run_flags_of_the_binding_are_the_header_s():
    """the BT_RUN_* values of include/bevy_terrain_amd.h, parsed from the text, against the constants the Python binding passes"""
    import re

    from bevy_terrain_amd import _ffi

    text = open(_ffi.HEADER_PATH).read()
    header = {m.group(1): int(m.group(2)) for m in re.finditer(r"\bBT_RUN_([A-Z_]+)\s*=\s*(\d+)", text)}
    assert len(header) == 10 and header["REFERENCE_DISPATCH"] == 256
    for name, value in header.items():
        assert getattr(_ffi, "RUN_" + name) == value, name


def _rust_layout(rust_text):
    """Parse integration/hip.rs back: {struct: [(field, offset, size)], "__size__"} by the repr(C) rules, the extern fn signatures and the
    constants — written against the generated text only (no import of the generator's parser)."""
    import re

    scalars = {"u8": (1, 1), "i8": (1, 1), "u16": (2, 2), "i16": (2, 2), "u32": (4, 4), "i32": (4, 4), "f32": (4, 4), "u64": (8, 8), "i64": (8, 8), "f64": (8, 8),
               "usize": (8, 8), "c_char": (1, 1), "bt_status": (4, 4)}
    structs, sizes = {}, {}

    def size_align(t):
        t = t.strip()
        if t.startswith("*"):
            return 8, 8
        m = re.fullmatch(r"\[(.+);\s*(\d+)\]", t)
        if m:
            s, a = size_align(m.group(1))
            return s * int(m.group(2)), a
        if t in scalars:
            return scalars[t]
        return sizes[t]

    for m in re.finditer(r"#\[repr\(C\)\]\s*(?:#\[derive\([^)]*\)\]\s*)?pub struct (\w+) \{(.*?)\n\}", rust_text, flags=re.S):
        name, body = m.group(1), m.group(2)
        fields, offset, align = [], 0, 1
        for fm in re.finditer(r"^\s*(?:pub )?(?:r#)?(\w+): (.+?),\s*$", body, flags=re.M):
            fname, ftype = fm.group(1), fm.group(2)
            s, a = size_align(ftype)
            offset = (offset + a - 1) // a * a
            fields.append((fname, offset, s))
            offset += s
            align = max(align, a)
        sizes[name] = ((offset + align - 1) // align * align, align)
        structs[name] = fields
    block = re.search(r'extern "C" \{(.*?)\n\}', rust_text, flags=re.S).group(1)
    fns = {}
    for fm in re.finditer(r"pub fn (\w+)\((.*?)\)( -> [^;]+)?;", block):
        args = [a for a in fm.group(2).split(", ") if a.strip()]
        fns[fm.group(1)] = (len(args), (fm.group(3) or "").replace(" -> ", ""))
    consts = {m.group(1): int(m.group(2), 0) for m in re.finditer(r"pub const (\w+): \w+ = (-?(?:0x[0-9a-f]+|\d+));", rust_text)}
    return structs, sizes, fns, consts


def test_generated_rust_binding_is_complete_and_current():
    """integration/hip.rs (tools/gen_rust_ffi.py) is the WHOLE boundary as Rust: regenerating it from the header gives the committed file;
    parsed back it declares exactly the header's functions with the argument counts of the ctypes prototypes, every #[repr(C)] struct has
    the size and the field offsets of the ctypes mirror, the opaque handles are zero-sized, and the constants carry the header's values."""
    import ctypes as C
    import importlib.util

    spec = importlib.util.spec_from_file_location("gen_rust_ffi", os.path.join(ROOT, "tools", "gen_rust_ffi.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    committed = open(os.path.join(ROOT, "integration", "hip.rs")).read()
    assert gen.generate(open(_ffi.HEADER_PATH).read()) == committed, "integration/hip.rs is stale: python tools/gen_rust_ffi.py"
    structs, sizes, fns, consts = _rust_layout(committed)
    assert set(fns) == _ffi.header_symbols() == set(_ffi.PROTOTYPES)
    for name, (restype, argtypes) in _ffi.PROTOTYPES.items():
        assert fns[name][0] == len(argtypes), (name, fns[name], len(argtypes))
        assert (fns[name][1] == "") == (restype is None), (name, fns[name])
    mirror = {"bt_tile_coordinate": _ffi.TileCoordinateC, "bt_atlas_tile": _ffi.AtlasTileC, "bt_attachment_config": _ffi.AttachmentConfigC,
              "bt_terrain_config": _ffi.TerrainConfigC, "bt_raster": _ffi.RasterC, "bt_preprocess_dataset": _ffi.PreprocessDatasetC,
              "bt_spherical_dataset": _ffi.SphericalDatasetC, "bt_tile_tree_entry": _ffi.TileTreeEntryC, "bt_run_stats": _ffi.RunStatsC,
              "bt_stream_stats": _ffi.StreamStatsC, "bt_shard_range": _ffi.ShardRangeC, "bt_launch_profile": _ffi.LaunchProfileC,
              "bt_side_parameter": _ffi.SideParameterC, "bt_view_state": _ffi.ViewStateC, "bt_indirect": _ffi.IndirectC,
              "bt_terrain_model": _ffi.TerrainModelC, "bt_terrain_view_config": _ffi.TerrainViewConfigC}
    checked = 0
    for name, cls in mirror.items():
        assert sizes[name][0] == C.sizeof(cls), (name, sizes[name], C.sizeof(cls))
        assert [f[0] for f in structs[name]] == [f[0] for f in cls._fields_], name
        for fname, offset, size in structs[name]:
            d = getattr(cls, fname)
            assert (d.offset, d.size) == (offset, size), (name, fname)
            checked += 1
    assert checked > 80
    # every struct and handle of the header is there (mirrored in ctypes or not)
    import re

    header = re.sub(r"/\*.*?\*/", "", open(_ffi.HEADER_PATH).read(), flags=re.S)
    declared = set(re.findall(r"\}\s*(bt_\w+)\s*;", header))
    opaque = set(re.findall(r"typedef\s+struct\s+(bt_\w+)\s+\1\s*;", header))
    assert declared | opaque == set(structs), (declared | opaque) ^ set(structs)
    assert all(sizes[o][0] == 0 for o in opaque) and len(opaque) == 6
    assert sizes["bt_tile_coordinate"][0] == 16 and sizes["bt_atlas_tile"][0] == 32 and sizes["bt_indirect"][0] == 16  # the reference's GPU layouts
    assert consts["BT_ABI_VERSION"] == _ffi.header_abi_version() and consts["BT_ERR_ATLAS_OUT_OF_INDICES"] == -2
    assert consts["BT_RUN_REFERENCE_DISPATCH"] == _ffi.RUN_REFERENCE_DISPATCH and consts["BT_INVALID_ATLAS_INDEX"] == 0xFFFFFFFF


_NULL_SWEEP = r"""
import ctypes as C, sys
sys.path.insert(0, sys.argv[1])
from bevy_terrain_amd import _ffi
L = _ffi.lib()
for name in sorted(_ffi.PROTOTYPES):
    restype, argtypes = _ffi.PROTOTYPES[name]
    args = []
    for t in argtypes:
        if isinstance(t, type) and issubclass(t, C.Structure):
            args.append(t())
        elif t in (C.c_float, C.c_double):
            args.append(0.0)
        elif t in (C.c_char_p, C.c_void_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
            args.append(None)
        else:
            args.append(0)
    print("calling", name, flush=True)
    r = getattr(L, name)(*args)
    if restype is C.c_int32 and name.startswith(("bt_atlas_", "bt_preprocessor_", "bt_tile_tree_", "bt_tiling_prepass_", "bt_ctx_", "bt_comm_")) \
            and name not in ("bt_ctx_io_threads",):
        print("status", name, r, flush=True)
print("done", flush=True)
"""


def test_every_entry_point_survives_null_arguments():
    """A host binding's first bug is a NULL handle: every function of the ABI, called with all-zero arguments (NULL handles, NULL out
    pointers, zero counts), returns — an error status where it has one — instead of dereferencing.  No GPU is touched: the argument
    checks come before the first HIP call."""
    p = subprocess.run([os.sys.executable, "-c", _NULL_SWEEP, ROOT], capture_output=True, text=True, timeout=300)
    lines = p.stdout.strip().splitlines()
    assert p.returncode == 0 and lines and lines[-1] == "done", (p.returncode, lines[-2:], p.stderr[-400:])
    assert sum(1 for l in lines if l.startswith("calling ")) == len(_ffi.PROTOTYPES)
    for l in lines:
        if l.startswith("status "):
            _, name, r = l.split()
            if name in ("bt_atlas_pending_loads", "bt_atlas_tiles", "bt_preprocessor_task_counts"):
                continue  # counts: 0 for no object
            assert int(r) < 0, l  # BT_ERR_*: never BT_OK for a NULL handle


def test_tile_helpers_take_coordinates_that_are_not_tiles():
    """TileCoordinate::INVALID (coordinate.rs:156) and other non-tiles (a side past the cube's six, a LOD past 30) have no neighbours: the helpers
    answer INVALID for every slot instead of indexing the side tables with them."""
    import ctypes as C
    L = _ffi.lib()
    out = (_ffi.TileCoordinateC * 8)()
    for side, lod, x, y in [(0xFFFFFFFF,) * 4, (6, 3, 1, 1), (7, 0, 0, 0), (0, 31, 5, 5), (2, 40, 0, 0)]:
        for spherical in (0, 1):
            L.bt_tile_neighbours(_ffi.TileCoordinateC(side, lod, x, y), spherical, out)
            if lod > 30 or (spherical and side >= 6):
                assert all((o.side, o.lod, o.x, o.y) == (0xFFFFFFFF,) * 4 for o in out), (side, lod, spherical)
        L.bt_tile_children(_ffi.TileCoordinateC(side, lod, x, y), C.cast(out, C.POINTER(_ffi.TileCoordinateC)))
        buf = C.create_string_buffer(64)
        assert L.bt_tile_name(_ffi.TileCoordinateC(side, lod, x, y), buf, 64) > 0


def test_fused_direct_keeps_its_in_flight_registers_in_place(tmp_path):
    """fused_direct issues its source loads from inline assembly, two blocks ahead, and waits for them by hand-counted s_waitcnt: the compiler believes a
    loaded register holds its value from the load instruction on.  That is only true if it never COPIES such a register (a live-range split, a spill) between
    the load and the wait that covers it — a copy taken early is stale, and the load then lands in a register that may be somebody else's.  Round 6 met both
    (a 14-load variant for sources finer than the tile grid: one wave's block of wrong pixels in ~10^4 jobs, then deterministically with more registers tied;
    backed out, profiles/r06_gebco_size.txt).  This pins the shape of the compiled kernel: no scratch, and no register copy of a load destination in front of a wait."""
    src = os.path.join(ROOT, "bevy_terrain_amd", "csrc", "bt_fused.hip")
    out = str(tmp_path / "fused.s")
    cc = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950",
                         "-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S", "--cuda-device-only", "-o", out, src], capture_output=True, text=True)
    assert cc.returncode == 0, cc.stderr[-2000:]
    text = open(out).read()
    kernels = list(re.finditer(r"\n(_ZN2bt\S*fused_direct_rgba8_kernel\S*):.*?\.end_amdhsa_kernel", text, flags=re.S))
    assert len(kernels) == 3, "fused_direct_rgba8_kernel<false, false>, <true, false> and <false, true> expected in the ISA"
    for m in kernels:
        _check_in_flight_registers(m)


def _check_in_flight_registers(m):
    body = m.group(0).split("\n")
    assert re.search(r"\.amdhsa_private_segment_fixed_size 0\b", m.group(0)), "fused_direct spills to scratch: a spilled in-flight register is a stale one"
    dests, inside = set(), False
    for line in body:
        inside = "ASMSTART" in line or (inside and "ASMEND" not in line)
        mm = re.match(r"\s*global_load_dword (v\d+),", line)
        if inside and mm:
            dests.add(mm.group(1))
    assert len(dests) == 20, sorted(dests)  # two sets of (4 rows + 1) x 2 columns
    waits = [i for i, line in enumerate(body) if re.match(r"\s*s_waitcnt vmcnt\(\d+\)\s*$", line) and "ASMSTART" in body[i - 1]]
    assert len(waits) >= 10
    copies = []
    for w in waits:
        for line in body[max(0, w - 8):w - 1]:
            mm = re.match(r"\s*v_mov_b(32|64)_e32\s+\S+,\s*(v\d+|v\[\d+:\d+\])", line)
            if mm:
                regs = [mm.group(2)] if "[" not in mm.group(2) else [f"v{k}" for k in range(int(mm.group(2)[2:].split(":")[0]), int(mm.group(2).split(":")[1][:-1]) + 1)]
                if any(r in dests for r in regs):
                    copies.append(line.strip())
    assert not copies, f"load destinations are copied in front of a hand-counted wait: {copies[:6]}"
