"""ctypes binding of oracle/_ref/libbt_wgslref.so — the reference's own WGSL executed on the CPU.

TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py --verify).  The library is generated from the
UNMODIFIED shader files under /root/reference/src/shaders by oracle/wgsl_ref (wgsl2cpp.py + ref_harness.cpp); it can be
(re)built only where /root/reference exists (this container).  The GPU box receives the built .so with the snapshot and
never reads /root/reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import _oracle as O

REF_DIR = os.path.join(O.ORACLE_DIR, "wgsl_ref")
LIB_PATH = os.path.join(O.ORACLE_DIR, "_ref", "libbt_wgslref.so")
REFERENCE = "/root/reference"

_lib = None


def build(force: bool = False) -> str | None:
    """make -C oracle/wgsl_ref when the reference is present; returns the library path or None."""
    if os.path.isdir(os.path.join(REFERENCE, "src", "shaders")):
        subprocess.check_call(["make", "-C", REF_DIR, "-s"] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH if os.path.exists(LIB_PATH) else None


def available() -> bool:
    return build() is not None


def lib():
    global _lib
    if _lib is None:
        path = build()
        if path is None:
            raise RuntimeError("oracle/_ref/libbt_wgslref.so is missing and /root/reference is not here to build it")
        L = C.CDLL(path)
        L.wref_run_task.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.wref_run_task.restype = None
        L.wref_refine.argtypes = [C.POINTER(O.View), C.POINTER(O.Coord), C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        L.wref_refine.restype = C.c_long
        L.wref_should_be_divided.argtypes = [C.POINTER(O.View), O.Coord, C.POINTER(C.c_float)]
        L.wref_should_be_divided.restype = C.c_int
        L.wref_sources.restype = C.c_char_p
        _lib = L
    return _lib


def sources() -> list[str]:
    """'sha256  path' of every shader file the library was generated from"""
    return [s for s in lib().wref_sources().decode().split(";") if s]


def attach(atlas: "O.OracleAtlas") -> "O.OracleAtlas":
    """Route every Split / Downsample / Stitch task of this oracle atlas through the executed WGSL.  The queue, the atlas
    index allocation and the storage stay the oracle's (they restate the reference's Rust side)."""
    O.lib().orc_set_task_backend.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    O.lib().orc_set_task_backend.restype = None
    O.lib().orc_set_task_backend(atlas._h, C.cast(lib().wref_run_task, C.c_void_p), None)
    return atlas


def refine(view: "O.View", cap: int | None = None):
    """final tile list (append order of a sequential run), indirect args, per-pass tile counts — like O.refine"""
    cap = cap or view.tile_count
    out = (O.Coord * cap)()
    indirect = (C.c_uint32 * 4)()
    passes = (C.c_uint32 * (view.refinement_count + 1))()
    n = lib().wref_refine(C.byref(view), out, cap, indirect, passes)
    if n < 0:
        raise OverflowError("tile buffers overflowed")
    return [out[i].tuple() for i in range(n)], list(indirect), list(passes)


def should_be_divided(view: "O.View", tile):
    d = C.c_float()
    r = lib().wref_should_be_divided(C.byref(view), O.Coord(*tile), C.byref(d))
    return bool(r), d.value
