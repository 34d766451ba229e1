"""Stand-ins for the reference's datasets (assets/terrains/*/source/*: Gaia / GEBCO files that are not part of either repository): when a
source file of an example is missing it is synthesised — fBm height on the GPU (bt_synth_fbm_r16), a colour ramp of it for the albedo — and
written where the example expects it, as the 16-bit PNG / TIFF or 8-bit PNG the reference's examples load."""
import os

import numpy as np


def _write_png(path, array):
    """a minimal PNG writer (8-bit RGBA or 16-bit gray, filter 0): what the library's decoder reads back"""
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    h, w = array.shape[:2]
    if array.dtype == np.uint16:
        rows, bits, colour = array.astype(">u2").reshape(h, -1).view(np.uint8), 16, 0
    else:
        rows, bits, colour = array.reshape(h, -1), 8, 6
    raw = np.concatenate([np.zeros((h, 1), np.uint8), rows], axis=1).tobytes()
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, bits, colour, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 1)) + chunk(b"IEND", b""))


def _write_tiff(path, array):
    """a minimal little-endian TIFF writer (16-bit gray, one uncompressed strip)"""
    import struct

    h, w = array.shape
    data = array.astype("<u2").tobytes()
    entries = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8), (277, 3, 1, 1), (278, 4, 1, h), (279, 4, 1, len(data))]
    ifd = struct.pack("<H", len(entries))
    for tag, typ, count, value in entries:
        ifd += struct.pack("<HHI", tag, typ, count) + (struct.pack("<HH", value, 0) if typ == 3 else struct.pack("<I", value))
    ifd += struct.pack("<I", 0)
    with open(path, "wb") as f:
        f.write(b"II*\x00" + struct.pack("<I", 8 + len(data)) + data + ifd)


def height(device, size, seed):
    ptr = device.synth_fbm_r16(size, size, seed)
    a = device.download(ptr, (size, size), np.uint16)
    device.free(ptr)
    return a


def albedo_of(h):
    t = (h >> 8).astype(np.uint8)
    return np.stack([np.maximum(t, 1), 255 - t // 2, 64 + t // 3, np.full_like(t, 255)], axis=2)


def ensure(path, make, log=print):
    """`path` exists afterwards; `make()` -> array is only called when it does not"""
    if os.path.exists(path):
        return
    os.makedirs(os.path.dirname(path), exist_ok=True)
    a = make()
    (_write_tiff if path.endswith((".tif", ".tiff")) else _write_png)(path, a)
    log(f"  (synthesised {path}: {a.shape[1]} x {a.shape[0]} {a.dtype})")
