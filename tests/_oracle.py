"""ctypes binding of the CPU oracle (oracle/libbt_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATH = os.path.join(ORACLE_DIR, "libbt_oracle.so")

INVALID = 0xFFFFFFFF
FORMAT_RGBA8 = 0
FORMAT_R16 = 1


class Coord(C.Structure):
    _fields_ = [("side", C.c_uint32), ("lod", C.c_uint32), ("x", C.c_uint32), ("y", C.c_uint32)]

    def tuple(self):
        return (self.side, self.lod, self.x, self.y)


class AttachmentConfig(C.Structure):
    _fields_ = [
        ("texture_size", C.c_uint32),
        ("border_size", C.c_uint32),
        ("mip_level_count", C.c_uint32),
        ("format", C.c_uint32),
    ]


class Dataset(C.Structure):
    _fields_ = [
        ("attachment_index", C.c_uint32),
        ("side", C.c_uint32),
        ("top_left", C.c_float * 2),
        ("bottom_right", C.c_float * 2),
        ("lod_begin", C.c_uint32),
        ("lod_end", C.c_uint32),
    ]


class SideParameter(C.Structure):
    _fields_ = [("view_xy", C.c_int32 * 2), ("view_uv", C.c_float * 2)]


class View(C.Structure):
    _fields_ = [
        ("spherical", C.c_uint32),
        ("tile_count", C.c_uint32),
        ("refinement_count", C.c_uint32),
        ("vertices_per_tile", C.c_uint32),
        ("subdivision_distance", C.c_float),
        ("origin_lod", C.c_uint32),
        ("approximate_height", C.c_float),
        ("sides", SideParameter * 6),
        ("world_position", C.c_float * 3),
        ("world_from_local", C.c_float * 12),
        ("local_from_world_transpose", C.c_float * 9),
    ]


class Model(C.Structure):
    _fields_ = [("kind", C.c_uint32), ("_pad", C.c_uint32), ("position", C.c_double * 3), ("a", C.c_double),
                ("b", C.c_double), ("min_height", C.c_float), ("max_height", C.c_float)]


class ViewConfig(C.Structure):
    _fields_ = [("tree_size", C.c_uint32), ("geometry_tile_count", C.c_uint32), ("refinement_count", C.c_uint32),
                ("grid_size", C.c_uint32), ("subdivision_tolerance", C.c_double),
                ("precision_threshold_distance", C.c_double), ("load_distance", C.c_double),
                ("morph_distance", C.c_double), ("blend_distance", C.c_double), ("morph_range", C.c_float),
                ("blend_range", C.c_float), ("origin_lod", C.c_uint32), ("_pad", C.c_uint32)]


class TreeEntry(C.Structure):
    _fields_ = [("atlas_index", C.c_uint32), ("atlas_lod", C.c_uint32)]


def usable_cores() -> int:
    """CPUs this process may really use: the affinity mask, capped by the cgroup CPU quota (a container sees all 256 hardware
    threads of the GPU box in os.cpu_count() but is throttled to its quota — more threads than that only buy stalls)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]  # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())  # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return max(1, n)


def build(force: bool = False) -> str:
    sources = [os.path.join(ORACLE_DIR, n) for n in ("bt_oracle.c", "bt_oracle_tree.c", "bt_oracle.h", "Makefile")]
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(p) > os.path.getmtime(_LIB_PATH) for p in sources if os.path.exists(p)
    )
    if force or stale:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    L = C.CDLL(build())
    vp, u32, i32, sz = C.c_void_p, C.c_uint32, C.c_int, C.c_size_t
    L.orc_children.argtypes = [Coord, C.POINTER(Coord)]
    L.orc_neighbours.argtypes = [Coord, i32, C.POINTER(Coord)]
    L.orc_parent.argtypes = [Coord]
    L.orc_parent.restype = Coord
    L.orc_atlas_new.argtypes = [u32, u32, i32, u32, C.POINTER(AttachmentConfig)]
    L.orc_atlas_new.restype = vp
    L.orc_atlas_free.argtypes = [vp]
    L.orc_clear_attachment.argtypes = [vp, u32]
    L.orc_preprocess_tile.argtypes = [vp, C.POINTER(Dataset), vp, u32, u32]
    L.orc_preprocess_spherical.argtypes = [vp, u32, u32, u32, C.POINTER(vp), u32, u32]
    L.orc_run.argtypes = [vp, i32]
    L.orc_task_count.argtypes = [vp, C.POINTER(u32)]
    L.orc_task_count.restype = u32
    L.orc_tile_count.argtypes = [vp]
    L.orc_tile_count.restype = u32
    L.orc_tiles.argtypes = [vp, C.POINTER(Coord), C.POINTER(u32), u32]
    L.orc_tiles.restype = u32
    L.orc_get_tile.argtypes = [vp, Coord]
    L.orc_get_tile.restype = u32
    L.orc_tile_data.argtypes = [vp, u32, u32]
    L.orc_tile_data.restype = vp
    L.orc_tile_bytes.argtypes = [vp, u32]
    L.orc_tile_bytes.restype = sz
    L.orc_save_attachment.argtypes = [vp, u32, C.c_char_p]
    L.orc_save_tile_config.argtypes = [vp, C.c_char_p]
    L.orc_tc_encode.argtypes = [C.POINTER(Coord), u32, vp, sz]
    L.orc_tc_encode.restype = sz
    L.orc_tc_decode.argtypes = [vp, sz, C.POINTER(Coord), u32]
    L.orc_tc_decode.restype = C.c_long
    L.orc_split_pixel.argtypes = [u32, u32, u32, Coord, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, u32,
                                  u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    L.orc_sample_tile.argtypes = [u32, u32, u32, vp, vp, vp]
    L.orc_sample_tile.restype = None
    L.orc_generate_mipmaps.argtypes = [u32, u32, u32, vp, vp]
    L.orc_generate_mipmaps.restype = sz
    L.orc_set_sampler_model.argtypes = [i32, i32]
    L.orc_set_sampler_model.restype = None
    dp = C.POINTER(C.c_double)
    L.orc_view_state_from_config.argtypes = [C.POINTER(Model), C.POINTER(ViewConfig), dp, C.c_float, C.POINTER(View)]
    L.orc_view_state_from_config.restype = None
    L.orc_coordinate_from_world_position.argtypes = [C.POINTER(Model), dp, dp]
    L.orc_coordinate_from_world_position.restype = u32
    L.orc_coordinate_world_position.argtypes = [C.POINTER(Model), u32, dp, C.c_float, dp]
    L.orc_coordinate_world_position.restype = None
    L.orc_project_point_ellipsoid.argtypes = [dp, dp, dp]
    L.orc_project_point_ellipsoid.restype = None
    L.orc_stream_new.argtypes = [u32, u32]
    L.orc_stream_new.restype = vp
    L.orc_stream_free.argtypes = [vp]
    L.orc_stream_add_existing.argtypes = [vp, C.POINTER(Coord), u32]
    L.orc_stream_request_tile.argtypes = [vp, Coord]
    L.orc_stream_release_tile.argtypes = [vp, Coord]
    L.orc_stream_get_best_tile.argtypes = [vp, Coord]
    L.orc_stream_get_best_tile.restype = TreeEntry
    L.orc_stream_pending_loads.argtypes = [vp]
    L.orc_stream_pending_loads.restype = u32
    L.orc_stream_finish_loads.argtypes = [vp, u32, C.POINTER(u32)]
    L.orc_stream_finish_loads.restype = u32
    L.orc_stream_atlas_index.argtypes = [vp, Coord]
    L.orc_stream_atlas_index.restype = u32
    L.orc_tile_tree_new.argtypes = [C.POINTER(Model), u32, C.POINTER(ViewConfig)]
    L.orc_tile_tree_new.restype = vp
    L.orc_tile_tree_free.argtypes = [vp]
    L.orc_tile_tree_update.argtypes = [vp, dp]
    L.orc_tile_tree_update.restype = None
    L.orc_tile_tree_released.argtypes = [vp, C.POINTER(Coord), u32]
    L.orc_tile_tree_released.restype = u32
    L.orc_tile_tree_requested.argtypes = [vp, C.POINTER(Coord), u32]
    L.orc_tile_tree_requested.restype = u32
    L.orc_tile_tree_apply_requests.argtypes = [vp, vp]
    L.orc_tile_tree_adjust_to_tile_atlas.argtypes = [vp, vp]
    L.orc_tile_tree_adjust_to_tile_atlas.restype = None
    L.orc_tile_tree_node_count.argtypes = [vp]
    L.orc_tile_tree_node_count.restype = u32
    L.orc_tile_tree_read.argtypes = [vp, vp, vp, vp, vp]
    L.orc_tile_tree_read.restype = None
    L.orc_tile_tree_set_approximate_height.argtypes = [vp, C.c_float]
    L.orc_tile_tree_set_approximate_height.restype = None
    L.orc_tile_tree_compute_blend.argtypes = [vp, dp, C.POINTER(u32), C.POINTER(C.c_float)]
    L.orc_tile_tree_compute_blend.restype = None
    L.orc_tile_tree_sample_attachment.argtypes = [vp, u32, u32, u32, C.POINTER(vp), u32, dp, u32, vp, vp]
    L.orc_tile_tree_sample_attachment.restype = None
    L.orc_refine.argtypes = [C.POINTER(View), C.POINTER(Coord), u32, C.POINTER(u32), C.POINTER(u32)]
    L.orc_refine.restype = C.c_long
    L.orc_should_be_divided.argtypes = [C.POINTER(View), Coord, C.POINTER(C.c_float)]
    L.orc_should_be_divided.restype = C.c_int
    _lib = L
    return L


def _np_ptr(a: np.ndarray):
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def children(c):
    out = (Coord * 4)()
    lib().orc_children(Coord(*c), out)
    return [o.tuple() for o in out]


def neighbours(c, spherical):
    out = (Coord * 8)()
    lib().orc_neighbours(Coord(*c), int(spherical), out)
    return [o.tuple() for o in out]


def texel_dtype(fmt):
    return np.uint16 if fmt == FORMAT_R16 else np.uint8


class OracleAtlas:
    """TileAtlas + Preprocessor of the oracle (one object, like the entity the reference spawns)."""

    def __init__(self, lod_count, atlas_size, spherical, attachments):
        """attachments: list of (texture_size, border_size, mip_level_count, format)."""
        self.attachments = list(attachments)
        cfg = (AttachmentConfig * len(attachments))(*[AttachmentConfig(*a) for a in attachments])
        self._h = lib().orc_atlas_new(lod_count, atlas_size, int(spherical), len(attachments), cfg)
        if not self._h:
            raise MemoryError("orc_atlas_new")
        self._keep = []

    def close(self):
        if self._h:
            lib().orc_atlas_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def clear_attachment(self, i):
        lib().orc_clear_attachment(self._h, i)
        return self

    def preprocess_tile(self, attachment_index, src, lod_range, side=0, top_left=(0.0, 0.0),
                        bottom_right=(1.0, 1.0)):
        src = np.ascontiguousarray(src)
        self._keep.append(src)
        d = Dataset(attachment_index, side, (C.c_float * 2)(*top_left), (C.c_float * 2)(*bottom_right),
                    lod_range[0], lod_range[1])
        rc = lib().orc_preprocess_tile(self._h, C.byref(d), _np_ptr(src), src.shape[1], src.shape[0])
        if rc:
            raise RuntimeError(f"orc_preprocess_tile rc={rc}")
        return self

    def preprocess_spherical(self, attachment_index, faces, lod_range):
        faces = [np.ascontiguousarray(f) for f in faces]
        self._keep.extend(faces)
        ptrs = (C.c_void_p * 6)(*[f.ctypes.data for f in faces])
        rc = lib().orc_preprocess_spherical(self._h, attachment_index, lod_range[0], lod_range[1], ptrs,
                                            faces[0].shape[1], faces[0].shape[0])
        if rc:
            raise RuntimeError(f"orc_preprocess_spherical rc={rc}")
        return self

    def task_counts(self):
        counts = (C.c_uint32 * 5)()
        n = lib().orc_task_count(self._h, counts)
        return n, dict(zip(("split", "stitch", "downsample", "save", "barrier"), list(counts)))

    def run(self, threads=1):
        rc = lib().orc_run(self._h, threads)
        if rc:
            raise RuntimeError(f"orc_run rc={rc}")
        return self

    def tiles(self):
        n = lib().orc_tile_count(self._h)
        coords = (Coord * max(n, 1))()
        idx = (C.c_uint32 * max(n, 1))()
        m = lib().orc_tiles(self._h, coords, idx, n)
        assert m == n
        return [(coords[i].tuple(), idx[i]) for i in range(n)]

    def get_tile(self, c):
        return lib().orc_get_tile(self._h, Coord(*c))

    def tile(self, attachment_index, atlas_index):
        T, _, _, fmt = self.attachments[attachment_index]
        nbytes = lib().orc_tile_bytes(self._h, attachment_index)
        p = lib().orc_tile_data(self._h, attachment_index, atlas_index)
        buf = (C.c_uint8 * nbytes).from_address(p)
        a = np.frombuffer(buf, dtype=texel_dtype(fmt)).copy()
        return a.reshape(T, T) if fmt == FORMAT_R16 else a.reshape(T, T, 4)

    def save_attachment(self, attachment_index, directory):
        rc = lib().orc_save_attachment(self._h, attachment_index, directory.encode())
        if rc:
            raise OSError(rc, "orc_save_attachment")

    def save_tile_config(self, path):
        rc = lib().orc_save_tile_config(self._h, path.encode())
        if rc:
            raise OSError(rc, "orc_save_tile_config")


class sampler_model:
    """with sampler_model(8): ... — the oracle's split filters with weights of that many fractional bits."""

    def __init__(self, fractional_bits, mode=0):
        self.args = (fractional_bits, mode)

    def __enter__(self):
        lib().orc_set_sampler_model(*self.args)

    def __exit__(self, *exc):
        lib().orc_set_sampler_model(0, 0)


def tc_encode(coords):
    arr = (Coord * max(len(coords), 1))(*[Coord(*c) for c in coords])
    n = lib().orc_tc_encode(arr, len(coords), None, 0)
    buf = (C.c_uint8 * n)()
    lib().orc_tc_encode(arr, len(coords), buf, n)
    return bytes(buf)


def tc_decode(data: bytes):
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    n = lib().orc_tc_decode(buf, len(data), None, 0)
    if n < 0:
        raise ValueError("bad tc")
    out = (Coord * max(n, 1))()
    lib().orc_tc_decode(buf, len(data), out, n)
    return [out[i].tuple() for i in range(n)]


def sample_tile(fmt, border_size, level0: np.ndarray, atlas_uv):
    level0 = np.ascontiguousarray(level0)
    uv = np.asarray(atlas_uv, dtype=np.float32)
    out = np.zeros(4, dtype=np.float32)
    lib().orc_sample_tile(fmt, level0.shape[0], border_size, _np_ptr(level0), _np_ptr(uv), _np_ptr(out))
    return out


def generate_mipmaps(fmt, level0: np.ndarray, mip_level_count: int):
    T = level0.shape[0]
    total = sum((T >> k) ** 2 for k in range(mip_level_count))
    ch = 1 if fmt == FORMAT_R16 else 4
    out = np.zeros(total * ch, dtype=texel_dtype(fmt))
    level0 = np.ascontiguousarray(level0)
    n = lib().orc_generate_mipmaps(fmt, T, mip_level_count, _np_ptr(level0), _np_ptr(out))
    assert n == total
    return out


def make_view(*, spherical, tile_count, refinement_count, vertices_per_tile, subdivision_distance, origin_lod,
              approximate_height, sides, world_position, world_from_local, local_from_world_transpose):
    v = View()
    v.spherical = int(spherical)
    v.tile_count = tile_count
    v.refinement_count = refinement_count
    v.vertices_per_tile = vertices_per_tile
    v.subdivision_distance = subdivision_distance
    v.origin_lod = origin_lod
    v.approximate_height = approximate_height
    for i, (xy, uv) in enumerate(sides):
        v.sides[i].view_xy[0], v.sides[i].view_xy[1] = int(xy[0]), int(xy[1])
        v.sides[i].view_uv[0], v.sides[i].view_uv[1] = float(uv[0]), float(uv[1])
    for i in range(3):
        v.world_position[i] = float(world_position[i])
    for i in range(12):
        v.world_from_local[i] = float(world_from_local[i])
    for i in range(9):
        v.local_from_world_transpose[i] = float(local_from_world_transpose[i])
    return v


def refine(view: View, cap=None):
    cap = cap or view.tile_count
    out = (Coord * cap)()
    indirect = (C.c_uint32 * 4)()
    passes = (C.c_uint32 * (view.refinement_count + 1))()
    n = lib().orc_refine(C.byref(view), out, cap, indirect, passes)
    if n < 0:
        raise OverflowError("orc_refine overflow")
    tiles = np.array([out[i].tuple() for i in range(n)], dtype=np.uint32).reshape(n, 4)
    return tiles, list(indirect), list(passes)


def should_be_divided(view: View, tile):
    d = C.c_float()
    r = lib().orc_should_be_divided(C.byref(view), Coord(*tile), C.byref(d))
    return bool(r), d.value


# ---- TileTree / streaming atlas / view-state derivation (oracle/bt_oracle_tree.c) -------------------------------

def make_model(kind, position, a, b=0.0, min_height=0.0, max_height=1.0):
    m = Model()
    m.kind = {"planar": 0, "spherical": 1, "ellipsoidal": 2}[kind]
    for i in range(3):
        m.position[i] = float(position[i])
    m.a, m.b, m.min_height, m.max_height = float(a), float(b), float(min_height), float(max_height)
    return m


def make_view_config(**kw):
    d = dict(tree_size=8, geometry_tile_count=1000000, refinement_count=30, grid_size=16, subdivision_tolerance=0.1,
             precision_threshold_distance=0.001, load_distance=2.5, morph_distance=16.0, blend_distance=2.0,
             morph_range=0.2, blend_range=0.2, origin_lod=10)
    d.update(kw)
    c = ViewConfig()
    for k, v in d.items():
        setattr(c, k, v)
    return c


def view_state_from_config(model, view_config, view_world_position, approximate_height):
    v = View()
    pos = (C.c_double * 3)(*view_world_position)
    lib().orc_view_state_from_config(C.byref(model), C.byref(view_config), pos, C.c_float(approximate_height), C.byref(v))
    return v


def coordinate_from_world_position(model, world):
    uv = (C.c_double * 2)()
    side = lib().orc_coordinate_from_world_position(C.byref(model), (C.c_double * 3)(*world), uv)
    return side, (uv[0], uv[1])


def coordinate_world_position(model, side, uv, height):
    out = (C.c_double * 3)()
    lib().orc_coordinate_world_position(C.byref(model), side, (C.c_double * 2)(*uv), C.c_float(height), out)
    return tuple(out)


def project_point_ellipsoid(e, y):
    out = (C.c_double * 3)()
    lib().orc_project_point_ellipsoid((C.c_double * 3)(*e), (C.c_double * 3)(*y), out)
    return tuple(out)


class Stream:
    """The streaming half of TileAtlasState."""

    def __init__(self, atlas_size, attachment_count, existing=()):
        self._h = lib().orc_stream_new(atlas_size, attachment_count)
        existing = list(existing)
        if existing:
            arr = (Coord * len(existing))(*[Coord(*c) for c in existing])
            lib().orc_stream_add_existing(self._h, arr, len(existing))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_stream_free(self._h)
            self._h = None

    def request_tile(self, c):
        return lib().orc_stream_request_tile(self._h, Coord(*c))

    def release_tile(self, c):
        return lib().orc_stream_release_tile(self._h, Coord(*c))

    def get_best_tile(self, c):
        e = lib().orc_stream_get_best_tile(self._h, Coord(*c))
        return e.atlas_index, e.atlas_lod

    def pending_loads(self):
        return lib().orc_stream_pending_loads(self._h)

    def finish_loads(self, n):
        out = (C.c_uint32 * (5 * max(n, 1)))()
        k = lib().orc_stream_finish_loads(self._h, n, out)
        return [((out[5 * i], out[5 * i + 1], out[5 * i + 2], out[5 * i + 3]), out[5 * i + 4]) for i in range(k)]

    def atlas_index(self, c):
        return lib().orc_stream_atlas_index(self._h, Coord(*c))


class TileTree:
    def __init__(self, model, lod_count, view_config):
        self.model, self.lod_count, self.view_config = model, lod_count, view_config
        self._h = lib().orc_tile_tree_new(C.byref(model), lod_count, C.byref(view_config))
        self.nodes = lib().orc_tile_tree_node_count(self._h)
        self.sides = 1 if model.kind == 0 else 6

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_tile_tree_free(self._h)
            self._h = None

    def update(self, view_position):
        lib().orc_tile_tree_update(self._h, (C.c_double * 3)(*view_position))
        cap = 2 * self.nodes + 16
        buf = (Coord * cap)()
        n = lib().orc_tile_tree_released(self._h, buf, cap)
        released = [buf[i].tuple() for i in range(n)]
        n = lib().orc_tile_tree_requested(self._h, buf, cap)
        requested = [buf[i].tuple() for i in range(n)]
        return released, requested

    def apply_requests(self, stream):
        rc = lib().orc_tile_tree_apply_requests(self._h, stream._h)
        if rc:
            raise RuntimeError(f"orc_tile_tree_apply_requests rc={rc}")

    def adjust_to_tile_atlas(self, stream):
        lib().orc_tile_tree_adjust_to_tile_atlas(self._h, stream._h)

    def read(self):
        entries = np.zeros((self.nodes, 2), np.uint32)
        origins = np.zeros((self.sides, self.lod_count, 2), np.uint32)
        coords = np.zeros((self.nodes, 4), np.uint32)
        requested = np.zeros(self.nodes, np.uint32)
        lib().orc_tile_tree_read(self._h, _np_ptr(entries), _np_ptr(origins), _np_ptr(coords), _np_ptr(requested))
        return entries, origins, coords, requested

    def set_approximate_height(self, h):
        lib().orc_tile_tree_set_approximate_height(self._h, C.c_float(h))

    def compute_blend(self, position):
        lod, ratio = C.c_uint32(), C.c_float()
        lib().orc_tile_tree_compute_blend(self._h, (C.c_double * 3)(*position), C.byref(lod), C.byref(ratio))
        return lod.value, ratio.value

    def sample_attachment(self, fmt, texture_size, border_size, layers, positions):
        """layers: {atlas_index: level-0 texel array}."""
        atlas_size = (max(layers) + 1) if layers else 1
        keep = {k: np.ascontiguousarray(v) for k, v in layers.items()}
        ptrs = (C.c_void_p * atlas_size)(*[keep[i].ctypes.data if i in keep else None for i in range(atlas_size)])
        positions = np.ascontiguousarray(positions, dtype=np.float64).reshape(-1, 3)
        n = len(positions)
        out = np.zeros((n, 4), np.float32)
        heights = np.zeros(n, np.float32)
        lib().orc_tile_tree_sample_attachment(self._h, fmt, texture_size, border_size, ptrs, atlas_size,
                                              positions.ctypes.data_as(C.POINTER(C.c_double)), n, _np_ptr(out), _np_ptr(heights))
        return out, heights
