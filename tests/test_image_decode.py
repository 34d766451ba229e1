"""the library's PNG / TIFF decoder (bt_image_load, host code) against Pillow's decode of the same
files — the source-image formats of the reference's examples (16-bit height PNG / TIFF, 8-bit albedo PNG)."""
import io
import os

import numpy as np
import pytest
from PIL import Image

import bevy_terrain_amd as bt
from bevy_terrain_amd.preprocess import decode_image

R16, RGBA8 = bt.AttachmentFormat.R16, bt.AttachmentFormat.Rgba8


def height(h=97, w=131, seed=1):
    rng = np.random.default_rng(seed)
    smooth = (np.add.outer(np.arange(h) * 311, np.arange(w) * 173) % 65536).astype(np.uint16)
    noise = rng.integers(0, 65536, size=(h, w), dtype=np.uint16)
    return np.where(rng.random((h, w)) < 0.5, smooth, noise)  # compressible runs + incompressible noise


def colour(h=75, w=90, channels=3, seed=2):
    rng = np.random.default_rng(seed)
    img = rng.integers(0, 256, size=(h, w, channels), dtype=np.uint8)
    img[20:40, :, :] = img[20:21, :1, :]  # flat band
    return img


def rgba_of(a):
    if a.ndim == 2:
        a = np.repeat(a[..., None], 3, axis=2)
    if a.shape[2] == 3:
        a = np.concatenate([a, np.full(a.shape[:2] + (1,), 255, np.uint8)], axis=2)
    return a


@pytest.mark.parametrize("compress_level", [0, 1, 6, 9])
def test_png_16_bit_gray(tmp_path, compress_level):
    a = height()
    p = str(tmp_path / "h.png")
    Image.fromarray(a).save(p, compress_level=compress_level)  # level 0: stored blocks; others: fixed / dynamic Huffman
    assert np.array_equal(decode_image(p, R16), a)
    assert np.array_equal(decode_image(open(p, "rb").read(), R16), a)


@pytest.mark.parametrize("mode", ["RGB", "RGBA", "L", "LA", "P"])
def test_png_8_bit_colour(tmp_path, mode):
    if mode == "RGB":
        src = colour(channels=3)
        im = Image.fromarray(src)
    elif mode == "RGBA":
        src = colour(channels=4)
        im = Image.fromarray(src)
    elif mode == "L":
        src = colour(channels=3)[..., 0]
        im = Image.fromarray(src)
    elif mode == "LA":  # gray + alpha: DynamicImage::into_rgba8 replicates the gray and keeps the alpha
        la = colour(channels=4)[..., :2]
        im = Image.fromarray(la, mode="LA")
        src = np.concatenate([np.repeat(la[..., :1], 3, axis=2), la[..., 1:]], axis=2)
    else:
        im = Image.fromarray(colour(channels=3)).quantize(colors=200)
        src = np.array(im.convert("RGB"))
    p = str(tmp_path / f"c_{mode}.png")
    im.save(p)
    assert np.array_equal(decode_image(p, RGBA8), rgba_of(src))


def test_png_all_filter_types_by_hand(tmp_path):
    """Pillow picks filters adaptively; this file forces each of the five filter types on its own rows."""
    import struct
    import zlib

    a = height(40, 33, seed=9)
    rows = a.astype(">u2").tobytes()
    stride, bpp = 33 * 2, 2
    raw = bytearray()
    prev = bytes(stride)
    for y in range(40):
        cur = rows[y * stride:(y + 1) * stride]
        f = y % 5
        out = bytearray(stride)
        for i in range(stride):
            A = cur[i - bpp] if i >= bpp else 0
            B = prev[i]
            Cc = prev[i - bpp] if i >= bpp else 0
            if f == 0:
                pred = 0
            elif f == 1:
                pred = A
            elif f == 2:
                pred = B
            elif f == 3:
                pred = (A + B) >> 1
            else:
                pp = A + B - Cc
                pa, pb, pc = abs(pp - A), abs(pp - B), abs(pp - Cc)
                pred = A if pa <= pb and pa <= pc else (B if pb <= pc else Cc)
            out[i] = (cur[i] - pred) & 0xFF
        raw += bytes([f]) + out
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    z = zlib.compress(bytes(raw), 6)
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 33, 40, 16, 0, 0, 0, 0)) + chunk(b"IDAT", z[:100]) + chunk(b"IDAT", z[100:]) + chunk(b"IEND", b"")
    assert np.array_equal(np.array(Image.open(io.BytesIO(png))), a)  # the hand-made file is a valid PNG
    assert np.array_equal(decode_image(png, R16), a)


@pytest.mark.parametrize("compression", [None, "tiff_lzw", "tiff_adobe_deflate", "packbits"])
def test_tiff_16_bit_gray(tmp_path, compression):
    a = height(203, 150, seed=4)
    p = str(tmp_path / "f.tif")
    Image.fromarray(a).save(p, **({"compression": compression} if compression else {}))
    assert np.array_equal(decode_image(p, R16), a)


def test_tiff_variants(tmp_path):
    a = height(130, 140, seed=5)
    # horizontal predictor + LZW (what GDAL writes for GEBCO-style rasters), several strips
    p = str(tmp_path / "pred.tif")
    Image.fromarray(a).save(p, compression="tiff_lzw", tiffinfo={317: 2, 278: 16})
    assert np.array_equal(np.array(Image.open(p)), a)
    assert np.array_equal(decode_image(p, R16), a)
    # 8-bit RGB, deflate
    c = colour(60, 70, 3)
    p = str(tmp_path / "rgb.tif")
    Image.fromarray(c).save(p, compression="tiff_adobe_deflate")
    assert np.array_equal(decode_image(p, RGBA8), rgba_of(c))
    # big-endian ("MM"), uncompressed, written by hand: header, image data, IFD
    import struct

    data = a.astype(">u2").tobytes()
    tags = [(256, 3, 1, 140), (257, 3, 1, 130), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, 8), (277, 3, 1, 1), (278, 3, 1, 130), (279, 4, 1, len(data))]
    ifd = struct.pack(">H", len(tags))
    for tag, typ, cnt, val in tags:
        ifd += struct.pack(">HHI", tag, typ, cnt) + (struct.pack(">HH", val, 0) if typ == 3 else struct.pack(">I", val))
    ifd += struct.pack(">I", 0)
    mm = b"MM" + struct.pack(">HI", 42, 8 + len(data)) + data + ifd
    assert np.array_equal(np.array(Image.open(io.BytesIO(mm))), a)
    assert np.array_equal(decode_image(mm, R16), a)
    # tiled, little-endian, by hand: 64 x 64 tiles over a 140 x 130 image (edge tiles padded)
    tw = th = 64
    across, down = 3, 3
    blob, offs, cnts = b"", [], []
    for ty in range(down):
        for tx in range(across):
            tile = np.zeros((th, tw), np.uint16)
            part = a[ty * th:(ty + 1) * th, tx * tw:(tx + 1) * tw]
            tile[:part.shape[0], :part.shape[1]] = part
            offs.append(8 + len(blob))
            cnts.append(tile.nbytes)
            blob += tile.astype("<u2").tobytes()
    extra = b"".join(struct.pack("<I", o) for o in offs) + b"".join(struct.pack("<I", c) for c in cnts)
    ifd_at = 8 + len(blob) + len(extra)
    tags = [(256, 3, 1, 140), (257, 3, 1, 130), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (277, 3, 1, 1), (322, 3, 1, tw), (323, 3, 1, th),
            (324, 4, 9, 8 + len(blob)), (325, 4, 9, 8 + len(blob) + 36)]
    ifd = struct.pack("<H", len(tags))
    for tag, typ, cnt, val in tags:
        ifd += struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 else struct.pack("<I", val))
    ifd += struct.pack("<I", 0)
    tiled = b"II" + struct.pack("<HI", 42, ifd_at) + blob + extra + ifd
    assert np.array_equal(np.array(Image.open(io.BytesIO(tiled))), a)
    assert np.array_equal(decode_image(tiled, R16), a)


def test_unsupported_and_corrupt_inputs_are_errors(tmp_path):
    a = height(20, 20)
    p = str(tmp_path / "h.png")
    Image.fromarray(a).save(p)
    with pytest.raises(bt._ffi.BtError) as e:
        decode_image(p, RGBA8)  # a 16-bit image is not an Rgba8 raster
    assert e.value.status == -5
    with pytest.raises(bt._ffi.BtError):
        decode_image(open(p, "rb").read()[:60], R16)  # truncated
    with pytest.raises(bt._ffi.BtError) as e:
        decode_image(b"GIF89a" + bytes(40), R16)
    assert e.value.status == -5
    with pytest.raises(bt._ffi.BtError) as e:
        decode_image(str(tmp_path / "missing.png"), R16)
    assert e.value.status == -4


def test_small_and_odd_images(tmp_path):
    one = np.array([[54321]], np.uint16)
    p = str(tmp_path / "one.png")
    Image.fromarray(one).save(p)
    assert np.array_equal(decode_image(p, R16), one)
    gray = np.arange(7 * 3, dtype=np.uint8).reshape(3, 7) * 11
    p = str(tmp_path / "g.tif")
    Image.fromarray(gray).save(p)
    assert np.array_equal(decode_image(p, RGBA8), rgba_of(gray))
    # a 16-bit RGB image fits neither attachment format
    import struct
    import zlib

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))

    raw = b"".join(b"\x00" + bytes(2 * 6 * 2) for _ in range(2))
    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 16, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b"")
    for fmt in (R16, RGBA8):
        with pytest.raises(bt._ffi.BtError) as e:
            decode_image(png, fmt)
        assert e.value.status == -5
    # an interlaced PNG is refused, not mis-decoded
    inter = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 2, 2, 8, 0, 0, 0, 1)) + chunk(b"IDAT", zlib.compress(b"\x00" * 12)) + chunk(b"IEND", b"")
    with pytest.raises(bt._ffi.BtError) as e:
        decode_image(inter, RGBA8)
    assert e.value.status == -5


def _tiff(entries, data=b"", big=False):
    """a classic TIFF with one IFD: entries = [(tag, type, count, value)] (SHORT = 3, LONG = 4; values inline), `data` appended"""
    import struct

    e = ">" if big else "<"
    out = (b"MM\x00\x2a" if big else b"II\x2a\x00") + struct.pack(e + "I", 8)
    out += struct.pack(e + "H", len(entries))
    for tag, typ, count, value in entries:
        out += struct.pack(e + "HHI", tag, typ, count)
        out += struct.pack(e + "HH", value, 0) if typ == 3 else struct.pack(e + "I", value)
    out += struct.pack(e + "I", 0)
    return out + data


def _png(width, height, idat, bits=16, color=0):
    import struct
    import zlib

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body) & 0xFFFFFFFF)

    return b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, bits, color, 0, 0, 0)) + chunk(b"IDAT", idat) + chunk(b"IEND", b"")


def test_hostile_headers_are_statuses_not_crashes():
    """every size a file states is validated before it sizes an allocation or a loop, the decompressors stop at
    the image's size, and nothing leaves bt_image_decode by exception"""
    import zlib

    base = 8 + 2 + 12 * 9 + 4  # where `data` starts in the 9-entry IFDs below
    good = [(256, 4, 1, 4), (257, 4, 1, 4), (258, 3, 1, 16), (259, 3, 1, 1), (262, 3, 1, 1), (273, 4, 1, base), (277, 3, 1, 1), (278, 4, 1, 4), (279, 4, 1, 32)]
    ok = _tiff(good, bytes(range(32)))
    assert decode_image(ok, R16).shape == (4, 4)

    def with_(tag, value, typ=4):
        return _tiff([(t, ty if t != tag else typ, c, v if t != tag else value) for t, ty, c, v in good], bytes(range(32)))

    for bad in (with_(278, 0),                       # RowsPerStrip = 0: used to divide by zero
                with_(256, 0x7FFFFFFF), with_(257, 0x7FFFFFFF),  # absurd dimensions
                with_(256, 1 << 21),
                _tiff(good[:-2] + [(322, 4, 1, 0x7FFFFFF0), (323, 4, 1, 0x7FFFFFF0), (324, 4, 1, base), (325, 4, 1, 32)][:2] + good[-2:], bytes(32)),  # TileWidth without offsets
                _tiff([e for e in good if e[0] not in (278,)] + [(322, 4, 1, 1 << 30), (323, 4, 1, 1 << 30)], bytes(32)),   # tile area wraps 32 bits
                _tiff([e for e in good if e[0] not in (278,)] + [(322, 4, 1, 16), (323, 4, 1, 0)], bytes(32))):              # TileLength = 0
        with pytest.raises(bt.BtError):
            decode_image(bad, R16)
    # PNG: dimensions beyond the cap; a zlib stream that inflates far beyond the image (cut at the image size: the decode
    # succeeds or fails, but the process neither aborts nor allocates the bomb)
    with pytest.raises(bt.BtError):
        decode_image(_png(1 << 24, 1 << 24, zlib.compress(b"\0" * 64)), R16)
    bomb = zlib.compress(b"\0" * (64 << 20), 9)  # 64 MiB of zeros in ~64 KB
    out = decode_image(_png(8, 8, bomb), R16)      # an 8 x 8 image: 8 * (1 + 16) bytes are taken, the rest is never produced
    assert out.shape == (8, 8) and not out.any()


def test_mutation_fuzz_under_address_sanitizer(tmp_path):
    """tests/fuzz/fuzz_image.cpp: the decoder's own source built with AddressSanitizer + UBSan, fed random mutants (flips, boundary values,
    truncations, duplicated / removed / spliced blocks) of every variant the decoder accepts, each decoded as R16 and as Rgba8 with every
    byte of a successful result read back.  Round 6 ran 24 M mutants clean (profiles/r06_hostile_inputs.txt); this keeps 60 000 in the suite."""
    import shutil
    import struct
    import subprocess
    import zlib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    clang = "/opt/rocm/lib/llvm/bin/clang++"
    if not os.path.exists(clang):
        clang = shutil.which("clang++") or shutil.which("g++")
    exe = str(tmp_path / "fuzz_image")
    build = subprocess.run([clang, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-D__HIP_PLATFORM_AMD__",
                            "-I/opt/rocm/include", "-I" + os.path.join(root, "include"), "-I" + os.path.join(root, "bevy_terrain_amd", "csrc"),
                            os.path.join(root, "tests", "fuzz", "fuzz_image.cpp"), os.path.join(root, "bevy_terrain_amd", "csrc", "bt_image.cpp"), "-o", exe],
                           capture_output=True, text=True)
    assert build.returncode == 0, build.stderr[-2000:]
    corpus = tmp_path / "corpus"
    corpus.mkdir()
    h, c = height(23, 17), colour(19, 13)
    Image.fromarray(h).save(str(corpus / "g16.png"))
    Image.fromarray(h).save(str(corpus / "g16_stored.png"), compress_level=0)
    Image.fromarray(h).save(str(corpus / "g16.tif"))
    for comp in ("tiff_lzw", "tiff_adobe_deflate", "packbits"):
        Image.fromarray(h).save(str(corpus / f"g16_{comp}.tif"), compression=comp)
    Image.fromarray(h).save(str(corpus / "g16_lzw_predictor.tif"), compression="tiff_lzw", tiffinfo={317: 2})
    Image.fromarray(c).save(str(corpus / "rgb.png"))
    Image.fromarray(c).save(str(corpus / "rgb.tif"))
    Image.fromarray(c).save(str(corpus / "rgb_lzw.tif"), compression="tiff_lzw")
    Image.fromarray(colour(19, 13, 4)).save(str(corpus / "rgba.png"))
    Image.fromarray(c[..., 0]).save(str(corpus / "g8.tif"))
    Image.fromarray(c).convert("P").save(str(corpus / "palette.png"))
    Image.fromarray(c[..., 0]).convert("LA").save(str(corpus / "gray_alpha.png"))

    def chunk(kind, body):
        return struct.pack(">I", len(body)) + kind + body + struct.pack(">I", zlib.crc32(kind + body))

    rows = np.random.default_rng(3).integers(0, 255, (5, 16), dtype=np.uint8)
    raw = b"".join(bytes([f]) + rows[f].tobytes() for f in range(5))  # one row per PNG filter type
    (corpus / "filters.png").write_bytes(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", 8, 5, 16, 0, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw)) + chunk(b"IEND", b""))
    seeds = sorted(str(p) for p in corpus.iterdir())
    for s in seeds:  # every seed is a file the decoder accepts (as one of the two formats)
        ok = 0
        for fmt in (R16, RGBA8):
            try:
                decode_image(s, fmt)
                ok += 1
            except bt.BtError:
                pass
        assert ok == 1, s
    run = subprocess.run([exe] + seeds + ["--", "60000", "7"], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
    assert run.returncode == 0 and run.stdout.startswith("iterations 60000"), (run.returncode, run.stdout[-300:], run.stderr[-3000:])
    assert int(run.stdout.split()[-1]) > 1000  # (a good share of the mutants still decode: the corpus reaches the pixel paths)
