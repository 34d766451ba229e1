// Source-image decode on the native side: what `asset_server.load(path)` + `preprocessor_load_tile`
// do for the reference's datasets (preprocess/preprocessor.rs:240, 401-422; formats/tiff.rs:14-62) — a 16-bit grayscale
// PNG or TIFF becomes the R16 raster of a height attachment, an 8-bit RGB(A) PNG / TIFF the Rgba8 raster of an albedo
// attachment (Bevy's Image::from_dynamic expands RGB to RGBA with alpha 255).  Third-party crates upstream (image 0.25,
// tiff 0.9, Cargo.toml:21-22); here: a self-contained inflate, PNG (all five filters, 8 / 16 bit, gray / RGB / RGBA /
// gray+alpha / palette, non-interlaced) and baseline TIFF (II / MM, strips or tiles, uncompressed / LZW / deflate /
// PackBits, horizontal predictor, chunky planar configuration).  Host code only; no device work.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <new>
#include <string>
#include <vector>

#include "bt_internal.hpp"

using namespace bt;

namespace {

// ------------------------------------------------------------------------------------------- inflate (RFC 1951)
struct BitReader {
    const uint8_t* p;
    size_t n, pos = 0;
    uint32_t bits = 0;
    int count = 0;
    bool fail = false;
    uint32_t get(int k) {  // k <= 16, LSB first
        while (count < k) {
            if (pos >= n) {
                fail = true;
                return 0;
            }
            bits |= uint32_t(p[pos++]) << count;
            count += 8;
        }
        const uint32_t v = bits & ((1u << k) - 1u);
        bits >>= k;
        count -= k;
        return v;
    }
    void align() {
        bits = 0;
        count = 0;
    }
};

struct Huffman {
    uint16_t count[16] = {}, symbol[320] = {};
    bool build(const uint8_t* lengths, int n) {
        memset(count, 0, sizeof count);
        for (int i = 0; i < n; i++) count[lengths[i]]++;
        count[0] = 0;
        uint16_t offs[16];
        offs[1] = 0;
        for (int i = 1; i < 15; i++) offs[i + 1] = offs[i] + count[i];
        for (int i = 0; i < n; i++)
            if (lengths[i]) symbol[offs[lengths[i]]++] = uint16_t(i);
        return true;
    }
    int decode(BitReader& br) const {
        int code = 0, first = 0, index = 0;
        for (int len = 1; len <= 15; len++) {
            code |= int(br.get(1));
            if (br.fail) return -1;
            const int c = count[len];
            if (code - c < first) return symbol[index + (code - first)];
            index += c;
            first += c;
            first <<= 1;
            code <<= 1;
        }
        return -1;
    }
};

bool inflate_raw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected) {
    static const uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
    static const uint16_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
    static const uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
    static const uint16_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
    BitReader br{src, n};
    if (expected) out.reserve(expected);
    // `expected` (when given) bounds the output: a stream that goes on past what the image holds is cut there — a hostile
    // file cannot make the decoder allocate more than the header-derived (and capped) image size
    const size_t limit = expected ? expected : size_t(-1);
    for (;;) {
        if (out.size() >= limit) return true;
        const uint32_t last = br.get(1), type = br.get(2);
        if (br.fail) return false;
        if (type == 0) {
            br.align();
            if (br.pos + 4 > n) return false;
            const uint32_t len = src[br.pos] | (src[br.pos + 1] << 8), nlen = src[br.pos + 2] | (src[br.pos + 3] << 8);
            br.pos += 4;
            if ((len ^ 0xFFFFu) != nlen || br.pos + len > n) return false;
            out.insert(out.end(), src + br.pos, src + br.pos + std::min<size_t>(len, limit - out.size()));
            br.pos += len;
        } else if (type == 1 || type == 2) {
            Huffman lit, dist;
            uint8_t lengths[320];
            if (type == 1) {
                for (int i = 0; i < 144; i++) lengths[i] = 8;
                for (int i = 144; i < 256; i++) lengths[i] = 9;
                for (int i = 256; i < 280; i++) lengths[i] = 7;
                for (int i = 280; i < 288; i++) lengths[i] = 8;
                lit.build(lengths, 288);
                for (int i = 0; i < 30; i++) lengths[i] = 5;
                dist.build(lengths, 30);
            } else {
                const int hlit = int(br.get(5)) + 257, hdist = int(br.get(5)) + 1, hclen = int(br.get(4)) + 4;
                static const uint8_t kOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {};
                for (int i = 0; i < hclen; i++) cl[kOrder[i]] = uint8_t(br.get(3));
                if (br.fail || hlit > 286 || hdist > 30) return false;
                Huffman clh;
                clh.build(cl, 19);
                int i = 0;
                while (i < hlit + hdist) {
                    const int sym = clh.decode(br);
                    if (sym < 0) return false;
                    if (sym < 16) {
                        lengths[i++] = uint8_t(sym);
                    } else {
                        int rep, val = 0;
                        if (sym == 16) {
                            if (i == 0) return false;
                            val = lengths[i - 1];
                            rep = 3 + int(br.get(2));
                        } else if (sym == 17) {
                            rep = 3 + int(br.get(3));
                        } else {
                            rep = 11 + int(br.get(7));
                        }
                        if (i + rep > hlit + hdist) return false;
                        while (rep--) lengths[i++] = uint8_t(val);
                    }
                }
                lit.build(lengths, hlit);
                dist.build(lengths + hlit, hdist);
            }
            for (;;) {
                const int sym = lit.decode(br);
                if (sym < 0) return false;
                if (out.size() >= limit) return true;
                if (sym < 256) {
                    out.push_back(uint8_t(sym));
                } else if (sym == 256) {
                    break;
                } else {
                    if (sym > 285) return false;
                    const uint32_t len = kLenBase[sym - 257] + br.get(kLenExtra[sym - 257]);
                    const int ds = dist.decode(br);
                    if (ds < 0 || ds > 29) return false;
                    const uint32_t d = kDistBase[ds] + br.get(kDistExtra[ds]);
                    if (br.fail || d > out.size()) return false;
                    size_t from = out.size() - d;
                    for (uint32_t k = 0; k < len && out.size() < limit; k++) out.push_back(out[from + k]);
                }
            }
        } else {
            return false;
        }
        if (last) return true;
    }
}

bool inflate_zlib(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected) {
    if (n < 6 || (src[0] & 0x0F) != 8 || ((src[0] << 8) | src[1]) % 31 != 0 || (src[1] & 0x20)) return false;
    return inflate_raw(src + 2, n - 2, out, expected);  // (the adler-32 trailer is not checked)
}

// ------------------------------------------------------------------------------------------- decoded image -> raster
// Header fields come from the file: everything derived from them is capped before it sizes an allocation or a loop
constexpr uint32_t kMaxSide = 1u << 20;
constexpr uint64_t kMaxImageBytes = 1ull << 34;  // 16 GiB decoded
inline bool dimensions_ok(uint64_t w, uint64_t h, uint64_t bytes_per_pixel) {
    return w != 0 && h != 0 && w <= kMaxSide && h <= kMaxSide && w * h * bytes_per_pixel <= kMaxImageBytes;
}

struct Decoded {
    uint32_t width = 0, height = 0, channels = 0, bits = 0;  // interleaved samples; 16-bit samples in HOST byte order
    std::vector<uint8_t> data;
};

bt_status to_raster(const Decoded& d, uint32_t format, bt_image* out) {
    const size_t pixels = size_t(d.width) * d.height;
    if (format == BT_FORMAT_R16) {
        if (d.bits != 16 || d.channels != 1) {
            set_error("image has %u channel(s) of %u bits; an R16 attachment needs one 16-bit channel", d.channels, d.bits);
            return BT_ERR_UNSUPPORTED;
        }
        uint8_t* buf = (uint8_t*)malloc(pixels * 2 ? pixels * 2 : 1);
        if (!buf) return BT_ERR_OUT_OF_MEMORY;
        memcpy(buf, d.data.data(), pixels * 2);
        *out = {buf, d.width, d.height, BT_FORMAT_R16, uint64_t(d.width) * 2};
        return BT_OK;
    }
    if (format == BT_FORMAT_RGBA8) {
        if (d.bits != 8 || d.channels == 0 || d.channels > 4) {
            set_error("image has %u channel(s) of %u bits; an Rgba8 attachment needs 8-bit gray, gray + alpha, RGB or RGBA", d.channels, d.bits);
            return BT_ERR_UNSUPPORTED;
        }
        uint8_t* buf = (uint8_t*)malloc(pixels * 4 ? pixels * 4 : 1);
        if (!buf) return BT_ERR_OUT_OF_MEMORY;
        const uint8_t* s = d.data.data();
        for (size_t i = 0; i < pixels; i++) {  // DynamicImage::into_rgba8: gray replicated, alpha 255
            if (d.channels == 1) {
                buf[4 * i] = buf[4 * i + 1] = buf[4 * i + 2] = s[i];
                buf[4 * i + 3] = 255;
            } else if (d.channels == 2) {  // LumaA8: gray replicated, its alpha kept
                buf[4 * i] = buf[4 * i + 1] = buf[4 * i + 2] = s[2 * i];
                buf[4 * i + 3] = s[2 * i + 1];
            } else {
                buf[4 * i] = s[d.channels * i];
                buf[4 * i + 1] = s[d.channels * i + 1];
                buf[4 * i + 2] = s[d.channels * i + 2];
                buf[4 * i + 3] = d.channels == 4 ? s[4 * i + 3] : 255;
            }
        }
        *out = {buf, d.width, d.height, BT_FORMAT_RGBA8, uint64_t(d.width) * 4};
        return BT_OK;
    }
    set_error("decode target format %u", format);
    return BT_ERR_UNSUPPORTED;
}

uint32_t be32(const uint8_t* p) { return (uint32_t(p[0]) << 24) | (uint32_t(p[1]) << 16) | (uint32_t(p[2]) << 8) | p[3]; }

// ------------------------------------------------------------------------------------------- PNG
bt_status decode_png(const uint8_t* p, size_t n, Decoded& d) {
    size_t pos = 8;
    uint32_t color = 0, interlace = 0;
    std::vector<uint8_t> idat, palette;
    bool have_header = false;
    while (pos + 12 <= n) {
        const uint32_t len = be32(p + pos);
        const uint8_t* type = p + pos + 4;
        if (pos + 12 + size_t(len) > n) break;
        const uint8_t* body = p + pos + 8;
        if (!memcmp(type, "IHDR", 4) && len >= 13) {
            d.width = be32(body);
            d.height = be32(body + 4);
            d.bits = body[8];
            color = body[9];
            interlace = body[12];
            have_header = true;
        } else if (!memcmp(type, "PLTE", 4)) {
            palette.assign(body, body + len);
        } else if (!memcmp(type, "IDAT", 4)) {
            idat.insert(idat.end(), body, body + len);
        } else if (!memcmp(type, "IEND", 4)) {
            break;
        }
        pos += 12 + size_t(len);
    }
    if (!have_header || d.width == 0 || d.height == 0) {
        set_error("PNG: no IHDR");
        return BT_ERR_IO;
    }
    if (!dimensions_ok(d.width, d.height, 8)) {
        set_error("PNG: %u x %u is beyond what this decoder accepts (%u per side, %llu bytes per image)", d.width, d.height, kMaxSide, (unsigned long long)kMaxImageBytes);
        return BT_ERR_UNSUPPORTED;
    }
    if (interlace) {
        set_error("PNG: Adam7 interlacing is not supported");
        return BT_ERR_UNSUPPORTED;
    }
    const uint32_t channels = color == 0 ? 1 : color == 2 ? 3 : color == 3 ? 1 : color == 4 ? 2 : color == 6 ? 4 : 0;
    if (!channels || (d.bits != 8 && d.bits != 16) || (color == 3 && d.bits != 8)) {
        set_error("PNG: colour type %u with %u bits per sample is not supported", color, d.bits);
        return BT_ERR_UNSUPPORTED;
    }
    const size_t bpp = size_t(channels) * d.bits / 8, stride = bpp * d.width;
    std::vector<uint8_t> raw;
    if (!inflate_zlib(idat.data(), idat.size(), raw, (stride + 1) * d.height) || raw.size() < (stride + 1) * d.height) {
        set_error("PNG: corrupt image data");
        return BT_ERR_IO;
    }
    std::vector<uint8_t> img(stride * d.height);
    for (uint32_t y = 0; y < d.height; y++) {  // unfilter (PNG spec 9.2)
        const uint8_t filter = raw[(stride + 1) * y];
        const uint8_t* in = &raw[(stride + 1) * y + 1];
        uint8_t* cur = &img[stride * y];
        const uint8_t* up = y ? &img[stride * (y - 1)] : nullptr;
        for (size_t i = 0; i < stride; i++) {
            const int a = i >= bpp ? cur[i - bpp] : 0, b = up ? up[i] : 0, c = (up && i >= bpp) ? up[i - bpp] : 0;
            int pred = 0;
            switch (filter) {
                case 0: pred = 0; break;
                case 1: pred = a; break;
                case 2: pred = b; break;
                case 3: pred = (a + b) >> 1; break;
                case 4: {
                    const int pp = a + b - c, pa = abs(pp - a), pb = abs(pp - b), pc = abs(pp - c);
                    pred = (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
                    break;
                }
                default:
                    set_error("PNG: filter type %u", filter);
                    return BT_ERR_IO;
            }
            cur[i] = uint8_t(in[i] + pred);
        }
    }
    if (color == 3) {  // palette -> RGB
        if (palette.size() < 3) {
            set_error("PNG: palette image without PLTE");
            return BT_ERR_IO;
        }
        d.data.resize(size_t(d.width) * d.height * 3);
        for (size_t i = 0; i < size_t(d.width) * d.height; i++) {
            const size_t e = std::min<size_t>(img[i], palette.size() / 3 - 1);
            memcpy(&d.data[3 * i], &palette[3 * e], 3);
        }
        d.channels = 3;
        return BT_OK;
    }
    if (d.bits == 16) {  // network byte order -> host
        uint16_t* w = (uint16_t*)img.data();
        for (size_t i = 0; i < img.size() / 2; i++) w[i] = uint16_t((img[2 * i] << 8) | img[2 * i + 1]);
    }
    d.channels = channels;
    d.data.swap(img);
    return BT_OK;
}

// ------------------------------------------------------------------------------------------- TIFF
struct TiffReader {
    const uint8_t* p;
    size_t n;
    bool big;
    uint16_t u16(size_t o) const { return o + 2 > n ? 0 : (big ? uint16_t((p[o] << 8) | p[o + 1]) : uint16_t(p[o] | (p[o + 1] << 8))); }
    uint32_t u32(size_t o) const {
        if (o + 4 > n) return 0;
        return big ? be32(p + o) : (uint32_t(p[o]) | (uint32_t(p[o + 1]) << 8) | (uint32_t(p[o + 2]) << 16) | (uint32_t(p[o + 3]) << 24));
    }
};

// TIFF 6.0 section 13: MSB-first codes, 9..12 bits, "early change"
bool tiff_lzw(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected) {
    struct Entry {
        uint16_t prev;
        uint8_t first, last;
        uint16_t length;
    };
    std::vector<Entry> table(4096);
    for (int i = 0; i < 256; i++) table[i] = {0xFFFF, uint8_t(i), uint8_t(i), 1};
    int next = 258, width = 9;
    uint32_t acc = 0;
    int nbits = 0;
    size_t pos = 0;
    int old = -1;
    out.reserve(expected);
    std::vector<uint8_t> tmp;
    auto emit = [&](int code) {
        const size_t len = table[code].length, base = out.size();
        out.resize(base + len);
        int c = code;
        for (size_t k = len; k-- > 0;) {
            out[base + k] = table[c].last;
            c = table[c].prev;
        }
    };
    for (;;) {
        while (nbits < width) {
            if (pos >= n) return out.size() >= expected;  // ran out of input without EOI: accept if complete
            acc = (acc << 8) | src[pos++];
            nbits += 8;
        }
        const int code = int((acc >> (nbits - width)) & ((1u << width) - 1u));
        nbits -= width;
        if (code == 257) return true;  // EOI
        if (code == 256) {             // clear
            next = 258;
            width = 9;
            old = -1;
            continue;
        }
        if (old < 0) {
            if (code >= 256) return false;
            emit(code);
            old = code;
            continue;
        }
        if (code < next) {
            emit(code);
            if (next < 4096) table[next++] = {uint16_t(old), table[old].first, table[code].first, uint16_t(table[old].length + 1)};
        } else if (code == next && next < 4096) {
            table[next++] = {uint16_t(old), table[old].first, table[old].first, uint16_t(table[old].length + 1)};
            emit(code);
        } else {
            return false;
        }
        old = code;
        if (next >= (1 << width) - 1 && width < 12) width++;  // early change
        if (out.size() >= expected && expected) return true;
    }
}

bool packbits(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t expected) {
    size_t pos = 0;
    while (pos < n && out.size() < expected) {
        const int8_t h = int8_t(src[pos++]);
        if (h >= 0) {
            const size_t len = size_t(h) + 1;
            if (pos + len > n) return false;
            out.insert(out.end(), src + pos, src + pos + len);
            pos += len;
        } else if (h != -128) {
            if (pos >= n) return false;
            out.insert(out.end(), size_t(1 - h), src[pos++]);
        }
    }
    return true;
}

bt_status decode_tiff(const uint8_t* p, size_t n, Decoded& d) {
    TiffReader r{p, n, p[0] == 'M'};
    if (r.u16(2) != 42) {
        set_error("TIFF: not a classic TIFF (BigTIFF is not supported)");
        return BT_ERR_UNSUPPORTED;
    }
    const size_t ifd = r.u32(4);
    const uint32_t entries = r.u16(ifd);
    if (!entries || ifd + 2 + size_t(entries) * 12 > n) {
        set_error("TIFF: bad directory");
        return BT_ERR_IO;
    }
    uint32_t compression = 1, photometric = 1, spp = 1, rows_per_strip = 0xFFFFFFFFu, planar = 1, predictor = 1, tile_w = 0, tile_h = 0, sample_format = 1;
    std::vector<uint32_t> offsets, counts;
    std::vector<uint32_t> bits_list;
    auto values = [&](size_t entry, std::vector<uint32_t>& v) {
        const uint32_t type = r.u16(entry + 2), cnt = r.u32(entry + 4);
        const uint32_t size = type == 3 ? 2 : (type == 4 ? 4 : (type == 1 ? 1 : 0));
        if (!size) return;
        size_t at = size_t(cnt) * size <= 4 ? entry + 8 : r.u32(entry + 8);
        v.clear();
        for (uint32_t i = 0; i < cnt && at + size <= n; i++, at += size) v.push_back(size == 2 ? r.u16(at) : (size == 4 ? r.u32(at) : p[at]));
    };
    for (uint32_t e = 0; e < entries; e++) {
        const size_t entry = ifd + 2 + size_t(e) * 12;
        const uint32_t tag = r.u16(entry);
        std::vector<uint32_t> v;
        values(entry, v);
        if (v.empty()) continue;
        switch (tag) {
            case 256: d.width = v[0]; break;
            case 257: d.height = v[0]; break;
            case 258: bits_list = v; break;
            case 259: compression = v[0]; break;
            case 262: photometric = v[0]; break;
            case 273: offsets = v; break;
            case 277: spp = v[0]; break;
            case 278: rows_per_strip = v[0]; break;
            case 279: counts = v; break;
            case 284: planar = v[0]; break;
            case 317: predictor = v[0]; break;
            case 322: tile_w = v[0]; break;
            case 323: tile_h = v[0]; break;
            case 324: offsets = v; break;
            case 325: counts = v; break;
            case 339: sample_format = v[0]; break;
            default: break;
        }
    }
    d.bits = bits_list.empty() ? 1 : bits_list[0];
    for (uint32_t b : bits_list)
        if (b != d.bits) d.bits = 0;
    if (!d.width || !d.height || (d.bits != 8 && d.bits != 16) || spp == 0 || spp > 4 || planar != 1 || photometric > 2 || sample_format > 2 ||
        offsets.empty() || offsets.size() != counts.size() || (predictor != 1 && predictor != 2)) {
        set_error("TIFF: unsupported layout (bits %u, samples %u, planar %u, photometric %u, predictor %u)", d.bits, spp, planar, photometric, predictor);
        return BT_ERR_UNSUPPORTED;
    }
    const size_t bps = d.bits / 8, px = bps * spp;
    if (!dimensions_ok(d.width, d.height, px)) {
        set_error("TIFF: %u x %u is beyond what this decoder accepts (%u per side, %llu bytes per image)", d.width, d.height, kMaxSide, (unsigned long long)kMaxImageBytes);
        return BT_ERR_UNSUPPORTED;
    }
    if ((tile_w != 0) != (tile_h != 0) || tile_w > kMaxSide || tile_h > kMaxSide || rows_per_strip == 0 ||
        (tile_w != 0 && uint64_t(tile_w) * tile_h * px > (1ull << 31))) {
        set_error("TIFF: bad strip / tile geometry (RowsPerStrip %u, TileWidth %u, TileLength %u)", rows_per_strip, tile_w, tile_h);
        return BT_ERR_IO;
    }
    const bool tiled = tile_w != 0 && tile_h != 0;
    const uint32_t cw = tiled ? tile_w : d.width, ch = tiled ? tile_h : std::min(rows_per_strip, d.height);
    const uint32_t across = tiled ? (d.width + cw - 1) / cw : 1, down = (d.height + ch - 1) / ch;
    if (size_t(across) * down > offsets.size()) {
        set_error("TIFF: %zu chunks listed, %zu needed", offsets.size(), size_t(across) * down);
        return BT_ERR_IO;
    }
    d.channels = spp;
    d.data.assign(size_t(d.width) * d.height * px, 0);
    std::vector<uint8_t> chunk;
    for (uint32_t cy = 0; cy < down; cy++)
        for (uint32_t cx = 0; cx < across; cx++) {
            const size_t i = size_t(cy) * across + cx;
            if (size_t(offsets[i]) + counts[i] > n) {
                set_error("TIFF: chunk %zu outside the file", i);
                return BT_ERR_IO;
            }
            const uint32_t rows = tiled ? ch : std::min(ch, d.height - cy * ch);
            const size_t expected = size_t(cw) * rows * px;  // <= 2^20 * 2^20 * 8: no wrap; strips are additionally bounded by the image cap
            if (expected == 0 || expected > kMaxImageBytes) {
                set_error("TIFF: chunk %zu has an impossible size", i);
                return BT_ERR_IO;
            }
            const uint8_t* src = p + offsets[i];
            chunk.clear();
            bool ok = true;
            if (compression == 1) chunk.assign(src, src + std::min<size_t>(counts[i], expected));
            else if (compression == 5) ok = tiff_lzw(src, counts[i], chunk, expected);
            else if (compression == 8 || compression == 32946) ok = inflate_zlib(src, counts[i], chunk, expected);
            else if (compression == 32773) ok = packbits(src, counts[i], chunk, expected);
            else {
                set_error("TIFF: compression %u is not supported", compression);
                return BT_ERR_UNSUPPORTED;
            }
            if (!ok || chunk.size() < expected) {
                set_error("TIFF: corrupt chunk %zu", i);
                return BT_ERR_IO;
            }
            // byte order of 16-bit samples -> host, then the horizontal predictor (per sample, per row)
            if (bps == 2) {
                uint16_t* w = (uint16_t*)chunk.data();
                for (size_t k = 0; k < expected / 2; k++) w[k] = r.big ? uint16_t((chunk[2 * k] << 8) | chunk[2 * k + 1]) : uint16_t(chunk[2 * k] | (chunk[2 * k + 1] << 8));
            }
            if (predictor == 2)
                for (uint32_t y = 0; y < rows; y++) {
                    if (bps == 2) {
                        uint16_t* row = (uint16_t*)&chunk[size_t(y) * cw * px];
                        for (size_t k = spp; k < size_t(cw) * spp; k++) row[k] = uint16_t(row[k] + row[k - spp]);
                    } else {
                        uint8_t* row = &chunk[size_t(y) * cw * px];
                        for (size_t k = spp; k < size_t(cw) * spp; k++) row[k] = uint8_t(row[k] + row[k - spp]);
                    }
                }
            const uint32_t x0 = cx * cw, y0 = cy * ch, copy_w = std::min(cw, d.width - x0);
            for (uint32_t y = 0; y < rows && y0 + y < d.height; y++)
                memcpy(&d.data[(size_t(y0 + y) * d.width + x0) * px], &chunk[size_t(y) * cw * px], size_t(copy_w) * px);
        }
    if (photometric == 0) {  // WhiteIsZero
        if (bps == 2) {
            uint16_t* w = (uint16_t*)d.data.data();
            for (size_t k = 0; k < d.data.size() / 2; k++) w[k] = uint16_t(~w[k]);
        } else {
            for (uint8_t& b : d.data) b = uint8_t(~b);
        }
    }
    return BT_OK;
}

}  // namespace

extern "C" {

bt_status bt_image_decode(const void* bytes, size_t n, uint32_t format, bt_image* out) {
    if (!bytes || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = {nullptr, 0, 0, 0, 0};
    const uint8_t* p = (const uint8_t*)bytes;
    static const uint8_t kPng[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1A, '\n'};
    // nothing may leave an extern "C" function by exception: a file that asks for more memory than there is becomes a status
    try {
        Decoded d;
        bt_status s;
        if (n >= 8 && !memcmp(p, kPng, 8)) s = decode_png(p, n, d);
        else if (n >= 8 && ((p[0] == 'I' && p[1] == 'I') || (p[0] == 'M' && p[1] == 'M'))) s = decode_tiff(p, n, d);
        else {
            set_error("neither a PNG nor a TIFF file");
            return BT_ERR_UNSUPPORTED;
        }
        if (s) return s;
        return to_raster(d, format, out);
    } catch (const std::bad_alloc&) {
        set_error("image decode: out of memory");
        return BT_ERR_OUT_OF_MEMORY;
    } catch (const std::exception& e) {
        set_error("image decode: %s", e.what());
        return BT_ERR_OUT_OF_MEMORY;
    }
}

bt_status bt_image_load(const char* path, uint32_t format, bt_image* out) {
    if (!path || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = {nullptr, 0, 0, 0, 0};
    FILE* f = fopen(path, "rb");
    if (!f) {
        set_error("cannot open %s", path);
        return BT_ERR_IO;
    }
    std::vector<uint8_t> buf;
    uint8_t tmp[1 << 16];
    size_t got;
    while ((got = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + got);
    fclose(f);
    return bt_image_decode(buf.data(), buf.size(), format, out);
}

void bt_image_free(bt_image* image) {
    if (!image) return;
    free(image->data);
    *image = {nullptr, 0, 0, 0, 0};
}

}  // extern "C"
