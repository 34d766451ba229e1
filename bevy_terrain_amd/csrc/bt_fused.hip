// Fused preprocess path: split + the LOD pyramid + border stitching in two kernel families.
//
// Why this is legal (verified by tests/test_oracle_preprocess.py against the oracle):
// the reference's split -> downsample -> stitch over per-tile tasks (preprocess/preprocessor.rs:234-343,
// shaders/preprocess/*.wgsl) equals "build, per LOD, the mosaic of tile centres; a tile is the window of
// its LOD's mosaic at stride c plus a b-pixel apron of the neighbours' centre pixels".  Every mosaic pixel
// is a pure function of its (tile, in-tile) coordinate, and a parent pixel is the valid-average of the
// 2x2 child-mosaic pixels below it (downsample.wgsl:18-20, child_size = c/2).  So nothing needs the
// reference's write-section copies, per-tile dispatches or phase barriers.
//
//   fused_main : one workgroup = 32 centre rows of one finest-LOD tile.  Streams the source raster once,
//                writes complete, already stitched 1024-byte tile rows of the finest LOD (aprons are pulled:
//                evaluated with the neighbour tile's own formula, so they are bit-identical to its centre),
//                and reduces in registers / across lane pairs to the next two LODs, which it *pushes* into
//                the parent and grand-parent tiles including the aprons of their neighbours.
//   fused_tail : the remaining (tiny) LODs, three at a time, from the atlas: 32x32 mosaic pixels per
//                workgroup, LDS hand-off between levels, same push.
//   cube seams : tiles on a cube-face edge get their aprons from the generic stitch kernel afterwards.
//
// Arithmetic contract: identical to bt_kernels.hip / oracle (IEEE binary32, -ffp-contract=off).
#include "bt_internal.hpp"

namespace bt {

namespace {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;
constexpr uint32_t kMainRows = 32;  // centre rows per fused_main workgroup (multiple of 4)

struct MainItem {  // one finest-LOD tile
    uint32_t side, x, y, atlas_index, raster;
};

struct FusedArgs {
    AttachmentMeta m;
    uint16_t* atlas;
    const RasterDev* rasters;
    const MainItem* items;
    const uint32_t* grids;         // per (side, lod): n x n atlas indices, x-major (x * n + y), INVALID = absent
    const uint32_t* grid_offsets;  // [side * 32 + lod] -> offset into grids, INVALID = no grid
    float tlx, tly, brx, bry;
    uint32_t lod;         // finest LOD of this launch (fused_main) / input LOD (fused_tail)
    uint32_t levels;      // LODs produced by this launch: main 1..3 (lod, lod-1, lod-2); tail 1..3 below lod
    uint32_t item_count;  // fused_main: tiles
    uint32_t groups;      // fused_main: row groups per tile
    uint32_t sides;       // fused_tail: 1 or 6
};

__device__ __forceinline__ float unorm16_to_float(uint32_t t) { return float(t) / 65535.0f; }
__device__ __forceinline__ uint32_t float_to_unorm16(float e) {
    const float cl = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
    return uint32_t(floorf(0.5f + 65535.0f * cl));
}
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

struct Axis {
    int i0, i1;
    float fr;
};

// split.wgsl:25-32 for centre coordinate r (0..c-1) of tile index `tile` (see bt_kernels.hip split_axis)
__device__ __forceinline__ Axis split_axis(uint32_t r, uint32_t c, uint32_t tile, float scale, float lo, float hi, uint32_t dim) {
    const float tc = float(r) / float(c);
    const float s = (float(tile) + tc) / scale;
    const float u = (s - lo) / (hi - lo);
    const float q = u * float(dim) - 0.5f;
    const float fl = floorf(q);
    Axis a;
    a.fr = q - fl;
    const int i = int(fl);
    const int last = int(dim) - 1;
    a.i0 = min(max(i, 0), last);
    a.i1 = min(max(i + 1, 0), last);
    return a;
}

__device__ __forceinline__ uint32_t grid_lookup(const FusedArgs& A, uint32_t side, uint32_t lod, int x, int y) {
    const int n = int(1u << lod);
    if (x < 0 || y < 0 || x >= n || y >= n) return kInvalid;
    const uint32_t off = A.grid_offsets[side * 32u + lod];
    if (off == kInvalid) return kInvalid;
    return A.grids[off + uint32_t(x) * uint32_t(n) + uint32_t(y)];
}

// A tile of some LOD together with its 8 same-face neighbours (region order of stitch.wgsl:57-66:
// N, E, S, W, NW, NE, SE, SW).  Out-of-face neighbours are treated as absent here; on a cube the
// face-edge tiles are re-stitched afterwards by the generic kernel.
struct TileNb {
    uint32_t self;
    uint32_t nb[8];
};

__device__ __forceinline__ TileNb load_tile_nb(const FusedArgs& A, uint32_t side, uint32_t lod, uint32_t x, uint32_t y) {
    constexpr int kOff[8][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    TileNb t;
    t.self = grid_lookup(A, side, lod, int(x), int(y));
#pragma unroll
    for (int r = 0; r < 8; r++) t.nb[r] = grid_lookup(A, side, lod, int(x) + kOff[r][0], int(y) + kOff[r][1]);
    return t;
}

// Write centre pixel (cx, cy) of tile `t` and every apron texel that copies it (stitch.wgsl:53-118 inverted):
//  - the tile's own apron where the neighbour on that side is absent (repeat_data clamps into the centre),
//  - the facing apron of each existing neighbour whose b-wide strip contains the pixel.
__device__ __forceinline__ void push_pixel(uint16_t* __restrict__ atlas, const TileNb& t, uint32_t T, uint32_t b, uint32_t c,
                                           uint32_t cx, uint32_t cy, uint16_t v) {
    const uint64_t tile_texels = uint64_t(T) * T;
    uint16_t* self = atlas + uint64_t(t.self) * tile_texels;
    self[uint64_t(b + cy) * T + b + cx] = v;
    const int ex = cx < b ? -1 : (cx >= c - b ? 1 : 0);
    const int ey = cy < b ? -1 : (cy >= c - b ? 1 : 0);
    if (ex == 0 && ey == 0) return;
    const uint32_t o = b + c;
    // own aprons (absent neighbour): only the outermost centre row / column / corner pixel is replicated
    const bool first_x = cx == 0, last_x = cx == c - 1, first_y = cy == 0, last_y = cy == c - 1;
    if (first_x && t.nb[3] == kInvalid)
        for (uint32_t j = 0; j < b; j++) self[uint64_t(b + cy) * T + j] = v;
    if (last_x && t.nb[1] == kInvalid)
        for (uint32_t j = 0; j < b; j++) self[uint64_t(b + cy) * T + o + j] = v;
    if (first_y && t.nb[0] == kInvalid)
        for (uint32_t j = 0; j < b; j++) self[uint64_t(j) * T + b + cx] = v;
    if (last_y && t.nb[2] == kInvalid)
        for (uint32_t j = 0; j < b; j++) self[uint64_t(o + j) * T + b + cx] = v;
    if ((first_x || last_x) && (first_y || last_y)) {
        const uint32_t region = first_y ? (first_x ? 4u : 5u) : (first_x ? 7u : 6u);
        if (t.nb[region] == kInvalid) {
            const uint32_t x0 = first_x ? 0u : o, y0 = first_y ? 0u : o;
            for (uint32_t j = 0; j < b; j++)
                for (uint32_t i = 0; i < b; i++) self[uint64_t(y0 + j) * T + x0 + i] = v;
        }
    }
    // neighbours' aprons: apron texel (px, py) of the neighbour at offset (dx, dy) copies our texel
    // (px - dx*... ) i.e. our centre (cx, cy) lands at texture (b + cx - dx*c, b + cy - dy*c) of that neighbour
    if (ex != 0) {
        const uint32_t n = t.nb[ex < 0 ? 3 : 1];
        if (n != kInvalid) atlas[uint64_t(n) * tile_texels + uint64_t(b + cy) * T + uint32_t(int(b + cx) - ex * int(c))] = v;
    }
    if (ey != 0) {
        const uint32_t n = t.nb[ey < 0 ? 0 : 2];
        if (n != kInvalid) atlas[uint64_t(n) * tile_texels + uint64_t(uint32_t(int(b + cy) - ey * int(c))) * T + b + cx] = v;
    }
    if (ex != 0 && ey != 0) {
        const uint32_t region = ey < 0 ? (ex < 0 ? 4u : 5u) : (ex < 0 ? 7u : 6u);
        const uint32_t n = t.nb[region];
        if (n != kInvalid)
            atlas[uint64_t(n) * tile_texels + uint64_t(uint32_t(int(b + cy) - ey * int(c))) * T + uint32_t(int(b + cx) - ex * int(c))] = v;
    }
}

// downsample.wgsl:25-39 on four texels in OFFSETS order (0,0),(0,1),(1,0),(1,1) of (dx, dy)
__device__ __forceinline__ uint32_t downsample4(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11) {
    float value = 0.0f, count = 0.0f;
    if (t00 != 0) { value += unorm16_to_float(t00); count += 1.0f; }
    if (t01 != 0) { value += unorm16_to_float(t01); count += 1.0f; }
    if (t10 != 0) { value += unorm16_to_float(t10); count += 1.0f; }
    if (t11 != 0) { value += unorm16_to_float(t11); count += 1.0f; }
    if (count == 0.0f) return 0;
    return float_to_unorm16(value / count);
}

// general (slow) evaluation of the finest-LOD mosaic pixel (tile, r) from the source; used for the
// b x b corner aprons only.  `home` = atlas texel holding that pixel (for the keep-previous rule).
__device__ uint32_t split_value_slow(const FusedArgs& A, const RasterDev& r, uint32_t tx, uint32_t rx, uint32_t ty, uint32_t ry,
                                     uint32_t home_index) {
    const float scale = float(1u << A.lod);
    const uint32_t c = A.m.center_size, b = A.m.border_size, T = A.m.texture_size;
    const Axis ax = split_axis(rx, c, tx, scale, A.tlx, A.brx, r.width);
    const Axis ay = split_axis(ry, c, ty, scale, A.tly, A.bry, r.height);
    const uint16_t* row0 = (const uint16_t*)((const uint8_t*)r.data + uint64_t(ay.i0) * r.pitch);
    const uint16_t* row1 = (const uint16_t*)((const uint8_t*)r.data + uint64_t(ay.i1) * r.pitch);
    const uint32_t t00 = row0[ax.i0], t10 = row0[ax.i1], t01 = row1[ax.i0], t11 = row1[ax.i1];
    if (t00 == 0 || t10 == 0 || t01 == 0 || t11 == 0) {
        if (home_index == kInvalid) return 0;
        return A.atlas[uint64_t(home_index) * T * T + uint64_t(b + ry) * T + b + rx];
    }
    const float top = mixf(unorm16_to_float(t00), unorm16_to_float(t10), ax.fr);
    const float bot = mixf(unorm16_to_float(t01), unorm16_to_float(t11), ax.fr);
    return float_to_unorm16(mixf(top, bot, ay.fr));
}

struct Texel4 {  // the four source texels a column pair needs from one source row, converted
    float a0, a1, b0, b1;
    uint32_t zero_mask;  // bit k set if texel k == 0
};

__device__ __forceinline__ Texel4 load_row(const uint8_t* __restrict__ data, uint64_t pitch, int y, int xa0, int xa1, int xb0, int xb1) {
    const uint16_t* row = (const uint16_t*)(data + uint64_t(y) * pitch);
    const uint32_t ta0 = row[xa0], ta1 = row[xa1], tb0 = row[xb0], tb1 = row[xb1];
    Texel4 r;
    r.a0 = unorm16_to_float(ta0);
    r.a1 = unorm16_to_float(ta1);
    r.b0 = unorm16_to_float(tb0);
    r.b1 = unorm16_to_float(tb1);
    r.zero_mask = (ta0 == 0 ? 1u : 0u) | (ta1 == 0 ? 2u : 0u) | (tb0 == 0 ? 4u : 0u) | (tb1 == 0 ? 8u : 0u);
    return r;
}

// blockIdx -> logical work id such that each XCD (blocks b, b+8, b+16, ... run on XCD b % 8) walks a
// contiguous range of work: neighbouring row groups / tiles then share source halos through one L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t total) {
    const uint32_t q = total / 8u, r = total % 8u, xcd = bid % 8u, i = bid / 8u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + i;
}

__global__ __launch_bounds__(256) void fused_main_kernel(FusedArgs A) {
    __shared__ int s_y0[kMainRows + 16], s_y1[kMainRows + 16];
    __shared__ float s_fy[kMainRows + 16];
    __shared__ uint32_t s_ty[kMainRows + 16], s_ry[kMainRows + 16];

    const uint32_t work = xcd_remap(blockIdx.x, gridDim.x);
    const MainItem it = A.items[work / A.groups];
    const uint32_t g = work % A.groups;
    const RasterDev raster = A.rasters[it.raster];
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size, o = b + c;
    const uint32_t tid = threadIdx.x;
    const float scale = float(1u << A.lod);
    const uint64_t tile_texels = uint64_t(T) * T;

    const TileNb t5 = load_tile_nb(A, it.side, A.lod, it.x, it.y);
    TileNb t4{}, t3{};
    if (A.levels >= 2) t4 = load_tile_nb(A, it.side, A.lod - 1, it.x >> 1, it.y >> 1);
    if (A.levels >= 3) t3 = load_tile_nb(A, it.side, A.lod - 2, it.x >> 2, it.y >> 2);

    // texture rows handled by this workgroup: its centre rows plus the apron rows of the tile's first / last group
    const uint32_t cr0 = g * kMainRows, cr1 = min(c, cr0 + kMainRows);
    const uint32_t py_begin = g == 0 ? 0u : b + cr0, py_end = cr1 == c ? T : b + cr1;
    const uint32_t nrows = py_end - py_begin;

    if (tid < nrows) {
        const uint32_t py = py_begin + tid;
        uint32_t ty, ry;
        if (py < b) {  // top apron: the north neighbour's last centre rows, or (absent) clamped into the own centre
            if (t5.nb[0] != kInvalid) { ty = it.y - 1; ry = c - b + py; } else { ty = it.y; ry = 0; }
        } else if (py >= o) {
            if (t5.nb[2] != kInvalid) { ty = it.y + 1; ry = py - o; } else { ty = it.y; ry = c - 1; }
        } else {
            ty = it.y;
            ry = py - b;
        }
        const Axis ay = split_axis(ry, c, ty, scale, A.tly, A.bry, raster.height);
        s_y0[tid] = ay.i0;
        s_y1[tid] = ay.i1;
        s_fy[tid] = ay.fr;
        s_ty[tid] = ty;
        s_ry[tid] = ry;
    }

    // column pair of this thread: centre pairs first so that lanes (2m, 2m+1) hold the two halves of one
    // pixel of the level two below; the b/2 right and b/2 left apron pairs come last
    const uint32_t half_c = c / 2, half_b = b / 2;
    enum { kCentre, kRight, kLeft, kIdle } role;
    uint32_t px0;  // texture column of the pair's first pixel
    uint32_t txc[2], rxc[2];
    if (tid < half_c) {
        role = kCentre;
        px0 = b + 2 * tid;
        txc[0] = txc[1] = it.x;
        rxc[0] = 2 * tid;
        rxc[1] = 2 * tid + 1;
    } else if (tid < half_c + half_b) {
        role = kRight;
        const uint32_t j = 2 * (tid - half_c);
        px0 = o + j;
        for (int k = 0; k < 2; k++) {
            if (t5.nb[1] != kInvalid) { txc[k] = it.x + 1; rxc[k] = j + k; } else { txc[k] = it.x; rxc[k] = c - 1; }
        }
    } else if (tid < half_c + 2 * half_b) {
        role = kLeft;
        const uint32_t j = 2 * (tid - half_c - half_b);
        px0 = j;
        for (int k = 0; k < 2; k++) {
            if (t5.nb[3] != kInvalid) { txc[k] = it.x - 1; rxc[k] = c - b + j + k; } else { txc[k] = it.x; rxc[k] = 0; }
        }
    } else {
        role = kIdle;
        px0 = 0;
        txc[0] = txc[1] = it.x;
        rxc[0] = rxc[1] = 0;
    }
    const Axis axa = split_axis(rxc[0], c, txc[0], scale, A.tlx, A.brx, raster.width);
    const Axis axb = split_axis(rxc[1], c, txc[1], scale, A.tlx, A.brx, raster.width);
    const float fxa = axa.fr, fxb = axb.fr;
    // atlas tile holding each column's pixels (keep-previous rule reads it when the source has no data)
    const uint32_t home_col = role == kRight && t5.nb[1] != kInvalid ? t5.nb[1]
                              : role == kLeft && t5.nb[3] != kInvalid ? t5.nb[3] : t5.self;

    __syncthreads();

    const uint8_t* data = (const uint8_t*)raster.data;
    uint16_t* tile5 = A.atlas + uint64_t(t5.self) * tile_texels;

    Texel4 prev{};
    int prev_y = -1;
    uint32_t even_a = 0, even_b = 0;  // finest values of the even row of the current row pair
    uint32_t q_even = 0;              // level-1 value of the even row pair of the current quad

    for (uint32_t j = 0; j < nrows; j++) {
        const uint32_t py = py_begin + j;
        const bool apron_row = py < b || py >= o;
        const int y0 = s_y0[j], y1 = s_y1[j];
        const float fy = s_fy[j];
        uint32_t va, vb;
        if (role == kIdle) {
            va = vb = 0;
        } else if (apron_row && role != kCentre) {
            // b x b corner: governed by the diagonal neighbour alone (stitch.wgsl:57-66, 105-118)
            const uint32_t region = py < b ? (role == kLeft ? 4u : 5u) : (role == kLeft ? 7u : 6u);
            const uint32_t n = t5.nb[region];
            uint32_t v[2];
            for (uint32_t k = 0; k < 2; k++) {
                uint32_t tx, rx, ty, ry, home;
                if (n != kInvalid) {
                    tx = role == kLeft ? it.x - 1 : it.x + 1;
                    rx = role == kLeft ? c - b + (px0 + k) : (px0 + k) - o;
                    ty = py < b ? it.y - 1 : it.y + 1;
                    ry = py < b ? c - b + py : py - o;
                    home = n;
                } else {
                    tx = it.x;
                    rx = role == kLeft ? 0u : c - 1;
                    ty = it.y;
                    ry = py < b ? 0u : c - 1;
                    home = t5.self;
                }
                v[k] = split_value_slow(A, raster, tx, rx, ty, ry, home);
            }
            va = v[0];
            vb = v[1];
        } else {
            const Texel4 top = (y0 == prev_y) ? prev : load_row(data, raster.pitch, y0, axa.i0, axa.i1, axb.i0, axb.i1);
            const Texel4 bot = (y1 == y0) ? top : load_row(data, raster.pitch, y1, axa.i0, axa.i1, axb.i0, axb.i1);
            prev = bot;
            prev_y = y1;
            const uint32_t zero = top.zero_mask | bot.zero_mask;
            {
                const float tp = mixf(top.a0, top.a1, fxa), bt_ = mixf(bot.a0, bot.a1, fxa);
                va = float_to_unorm16(mixf(tp, bt_, fy));
            }
            {
                const float tp = mixf(top.b0, top.b1, fxb), bt_ = mixf(bot.b0, bot.b1, fxb);
                vb = float_to_unorm16(mixf(tp, bt_, fy));
            }
            if (zero) {  // no data in the footprint: the pixel keeps its previous atlas value (split.wgsl:37-42)
                const uint32_t home = apron_row ? (py < b ? t5.nb[0] : t5.nb[2]) : home_col;
                const uint32_t hi = home == kInvalid ? t5.self : home;
                const uint16_t* hrow = A.atlas + uint64_t(hi) * tile_texels + uint64_t(b + s_ry[j]) * T + b;
                if (zero & 3u) va = hrow[rxc[0]];
                if (zero & 12u) vb = hrow[rxc[1]];
            }
        }
        if (role != kIdle) ((uint32_t*)tile5)[(uint64_t(py) * T + px0) / 2] = va | (vb << 16);

        // ---- next two LODs from the centre pixels: rows pair up, then lanes pair up
        if (A.levels >= 2 && !apron_row) {
            const uint32_t cy = py - b;
            if ((cy & 1u) == 0) {
                even_a = va;
                even_b = vb;
            } else {
                // level-1 pixel (tid, cy/2) of this tile's quadrant: OFFSETS (0,0),(0,1),(1,0),(1,1)
                const uint32_t q = downsample4(even_a, va, even_b, vb);
                if (role == kCentre)
                    push_pixel(A.atlas, t4, T, b, c, (it.x & 1u) * half_c + tid, (it.y & 1u) * half_c + (cy >> 1), uint16_t(q));
                if (A.levels >= 3) {
                    if (((cy >> 1) & 1u) == 0) {
                        q_even = q;
                    } else {
                        const uint32_t other_even = __shfl_xor(q_even, 1), other_odd = __shfl_xor(q, 1);
                        if (role == kCentre && (tid & 1u) == 0) {
                            const uint32_t w = downsample4(q_even, q, other_even, other_odd);
                            push_pixel(A.atlas, t3, T, b, c, (it.x & 3u) * (c / 4) + (tid >> 1), (it.y & 3u) * (c / 4) + (cy >> 2), uint16_t(w));
                        }
                    }
                }
            }
        }
    }
}

// ---- fused_tail: up to three LODs below `A.lod`, read from the atlas ---------------------------------
// Workgroup = 32 x 32 pixels of the LOD-`lod` mosaic (16 x 16 threads, 2 x 2 pixels each).
__global__ __launch_bounds__(256) void fused_tail_kernel(FusedArgs A) {
    __shared__ uint16_t s_l1[16][16];
    __shared__ uint16_t s_l2[8][8];
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size;
    const uint64_t tile_texels = uint64_t(T) * T;
    const uint32_t side = blockIdx.z;
    const uint32_t size = (1u << A.lod) * c;  // mosaic extent of the input LOD
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t gx = blockIdx.x * 32u + 2u * tx, gy = blockIdx.y * 32u + 2u * ty;  // first input pixel

    // level 1 (lod - 1): one pixel per thread
    uint32_t v1 = 0;
    const bool in1 = gx < size && gy < size;
    if (in1) {
        uint32_t t[4];
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) {  // (dx, dy) = (0,0),(0,1),(1,0),(1,1)
            const uint32_t x = gx + (k >> 1), y = gy + (k & 1u);
            const uint32_t idx = grid_lookup(A, side, A.lod, int(x / c), int(y / c));
            t[k] = idx == kInvalid ? 0u : A.atlas[uint64_t(idx) * tile_texels + uint64_t(b + y % c) * T + b + x % c];
        }
        v1 = downsample4(t[0], t[1], t[2], t[3]);
        const uint32_t x1 = gx >> 1, y1 = gy >> 1;
        const TileNb tn = load_tile_nb(A, side, A.lod - 1, x1 / c, y1 / c);
        if (tn.self != kInvalid) push_pixel(A.atlas, tn, T, b, c, x1 % c, y1 % c, uint16_t(v1));
    }
    if (A.levels < 2) return;
    s_l1[ty][tx] = uint16_t(v1);
    __syncthreads();
    // level 2: 8 x 8 per workgroup
    uint32_t v2 = 0;
    const uint32_t ux = threadIdx.x & 7u, uy = (threadIdx.x >> 3) & 7u;
    const bool act2 = threadIdx.x < 64 && (blockIdx.x * 32u + 4u * ux) < size && (blockIdx.y * 32u + 4u * uy) < size;
    if (act2) {
        v2 = downsample4(s_l1[2 * uy][2 * ux], s_l1[2 * uy + 1][2 * ux], s_l1[2 * uy][2 * ux + 1], s_l1[2 * uy + 1][2 * ux + 1]);
        const uint32_t x2 = blockIdx.x * 8u + ux, y2 = blockIdx.y * 8u + uy;
        const TileNb tn = load_tile_nb(A, side, A.lod - 2, x2 / c, y2 / c);
        if (tn.self != kInvalid) push_pixel(A.atlas, tn, T, b, c, x2 % c, y2 % c, uint16_t(v2));
    }
    if (A.levels < 3) return;
    if (threadIdx.x < 64) s_l2[uy][ux] = uint16_t(v2);
    __syncthreads();
    const uint32_t wx = threadIdx.x & 3u, wy = (threadIdx.x >> 2) & 3u;
    if (threadIdx.x < 16 && (blockIdx.x * 32u + 8u * wx) < size && (blockIdx.y * 32u + 8u * wy) < size) {
        const uint32_t v3 = downsample4(s_l2[2 * wy][2 * wx], s_l2[2 * wy + 1][2 * wx], s_l2[2 * wy][2 * wx + 1], s_l2[2 * wy + 1][2 * wx + 1]);
        const uint32_t x3 = blockIdx.x * 4u + wx, y3 = blockIdx.y * 4u + wy;
        const TileNb tn = load_tile_nb(A, side, A.lod - 3, x3 / c, y3 / c);
        if (tn.self != kInvalid) push_pixel(A.atlas, tn, T, b, c, x3 % c, y3 % c, uint16_t(v3));
    }
}

}  // namespace

// =============================================================================== host side: planning

struct FusedJobDev {  // device buffers of one fused job, kept alive in the preprocessor
    FusedArgs args;
    uint32_t attachment;
};

static std::vector<FusedJobDev>& jobs_of(bt_preprocessor* p);

}  // namespace bt

// storage for fused jobs lives next to the preprocessor; keyed by pointer to avoid widening the struct
#include <map>
#include <mutex>
namespace bt {
static std::mutex g_jobs_mutex;
static std::map<bt_preprocessor*, std::vector<FusedJobDev>> g_jobs;
static std::map<bt_preprocessor*, std::vector<void*>> g_job_allocs;

static std::vector<FusedJobDev>& jobs_of(bt_preprocessor* p) { return g_jobs[p]; }

void fused_release(bt_preprocessor* p) {
    std::lock_guard<std::mutex> lock(g_jobs_mutex);
    for (void* d : g_job_allocs[p]) hipFree(d);
    g_job_allocs.erase(p);
    g_jobs.erase(p);
}

template <typename V>
static bt_status upload_vector(bt_preprocessor* p, const std::vector<V>& v, const V** out) {
    void* d = nullptr;
    BT_HIP(hipMalloc(&d, v.size() * sizeof(V) ? v.size() * sizeof(V) : 1));
    g_job_allocs[p].push_back(d);
    if (!v.empty()) BT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice));
    *out = (const V*)d;
    return BT_OK;
}

// A job (one preprocess_tile / preprocess_spherical call) qualifies for the fused path when
//  - the attachment is R16 with T <= 512, even b, c % 4 == 0, c >= 2b;
//  - at every LOD but the finest, each queued tile has all four children queued (full quadtree below it).
// Otherwise the whole queue runs on the generic kernels.
bool fused_plan(bt_preprocessor* p, bt_atlas* a, std::vector<TaskDev>& tasks, std::vector<Launch>& plan) {
    std::lock_guard<std::mutex> lock(g_jobs_mutex);
    for (void* d : g_job_allocs[p]) hipFree(d);
    g_job_allocs[p].clear();
    std::vector<FusedJobDev>& jobs = jobs_of(p);
    jobs.clear();
    if (p->queue.empty()) return false;

    const std::vector<Task>& q = p->queue;
    for (uint32_t job = 0; job < p->jobs; job++) {
        // collect the job's tasks
        std::vector<const Task*> splits, downs, stitches;
        for (const Task& t : q) {
            if (t.job != job) continue;
            if (t.type == kSplit) splits.push_back(&t);
            else if (t.type == kDownsample) downs.push_back(&t);
            else if (t.type == kStitch) stitches.push_back(&t);
        }
        if (splits.empty()) return false;
        const uint32_t ai = splits[0]->attachment_index;
        const Attachment& at = a->attachments[ai];
        const AttachmentMeta& m = at.meta;
        if (m.format != BT_FORMAT_R16 || m.texture_size > 512 || (m.border_size & 1u) || (m.center_size & 3u) ||
            m.center_size < 2 * m.border_size || m.border_size == 0 || m.border_size > 8)
            return false;
        const uint32_t lod_hi = splits[0]->coord.lod;
        uint32_t lod_lo = lod_hi;
        for (const Task* t : downs) lod_lo = std::min(lod_lo, t->coord.lod);
        if (lod_hi > 24) return false;
        const bool spherical = a->config.spherical != 0;
        const uint32_t sides = spherical ? 6u : 1u;

        // grids per (side, lod)
        std::vector<uint32_t> grid_offsets(6 * 32, kInvalid), grids;
        for (uint32_t side = 0; side < sides; side++)
            for (uint32_t lod = lod_lo; lod <= lod_hi; lod++) {
                grid_offsets[side * 32 + lod] = uint32_t(grids.size());
                grids.resize(grids.size() + (size_t(1) << (2 * lod)), kInvalid);
            }
        auto cell = [&](const bt_tile_coordinate& c) -> uint32_t& {
            return grids[grid_offsets[c.side * 32 + c.lod] + (size_t(c.x) << c.lod) + c.y];
        };
        for (const Task* t : splits) {
            if (t->coord.side >= sides || t->coord.lod != lod_hi) return false;
            cell(t->coord) = t->atlas_index;
        }
        for (const Task* t : downs) {
            if (t->coord.side >= sides) return false;
            cell(t->coord) = t->atlas_index;
        }
        // completeness: every downsample tile has its four children in the grids
        for (const Task* t : downs) {
            bt_tile_coordinate ch[4];
            tile_children(t->coord, ch);
            for (int k = 0; k < 4; k++)
                if (cell(ch[k]) == kInvalid) return false;
        }
        // every tile that is stitched must be in the grids and vice versa (same tile set)
        size_t present = 0;
        for (uint32_t v : grids) present += v != kInvalid;
        if (present != stitches.size()) return false;
        // all split tasks share the dataset rectangle
        for (const Task* t : splits)
            if (t->tl[0] != splits[0]->tl[0] || t->tl[1] != splits[0]->tl[1] || t->br[0] != splits[0]->br[0] || t->br[1] != splits[0]->br[1])
                return false;

        std::vector<MainItem> items;
        for (const Task* t : splits) items.push_back({t->coord.side, t->coord.x, t->coord.y, t->atlas_index, uint32_t(t->raster)});

        FusedArgs args{};
        args.m = m;
        args.atlas = (uint16_t*)at.level0;
        args.rasters = p->rasters_dev;  // (re)allocated by bt_preprocessor_run before the first launch
        args.tlx = splits[0]->tl[0];
        args.tly = splits[0]->tl[1];
        args.brx = splits[0]->br[0];
        args.bry = splits[0]->br[1];
        args.sides = sides;
        if (upload_vector(p, items, &args.items) || upload_vector(p, grids, &args.grids) || upload_vector(p, grid_offsets, &args.grid_offsets))
            return false;

        const uint64_t bpp = 2, Tt = m.texture_size, cc = m.center_size;
        uint64_t source_bytes = 0;
        {
            std::vector<bool> seen(p->rasters.size(), false);
            for (const Task* t : splits)
                if (!seen[t->raster]) {
                    seen[t->raster] = true;
                    source_bytes += uint64_t(p->rasters[t->raster].dev.width) * p->rasters[t->raster].dev.height * bpp;
                }
        }
        auto tiles_at = [&](uint32_t lod) {
            uint64_t n = 0;
            for (uint32_t side = 0; side < sides; side++) {
                const uint32_t off = grid_offsets[side * 32 + lod];
                for (size_t i = 0; i < (size_t(1) << (2 * lod)); i++) n += grids[off + i] != kInvalid;
            }
            return n;
        };

        // main launch: finest LOD + up to two more
        const uint32_t nlods = lod_hi - lod_lo + 1;
        const uint32_t main_levels = std::min(3u, nlods);
        FusedJobDev main_job{args, ai};
        main_job.args.lod = lod_hi;
        main_job.args.levels = main_levels;
        main_job.args.item_count = uint32_t(items.size());
        main_job.args.groups = (m.center_size + kMainRows - 1) / kMainRows;
        Launch lm{};
        lm.kind = kLaunchFusedMain;
        lm.attachment = ai;
        lm.task_count = uint32_t(items.size());
        lm.aux0 = uint32_t(jobs.size());
        lm.algorithmic_bytes = source_bytes;
        for (uint32_t k = 0; k < main_levels; k++) lm.algorithmic_bytes += tiles_at(lod_hi - k) * Tt * Tt * bpp;
        jobs.push_back(main_job);
        plan.push_back(lm);

        // tail launches: three LODs at a time below the last fused one
        uint32_t in_lod = lod_hi - (main_levels - 1);
        while (in_lod > lod_lo) {
            const uint32_t levels = std::min(3u, in_lod - lod_lo);
            FusedJobDev tail{args, ai};
            tail.args.lod = in_lod;
            tail.args.levels = levels;
            Launch lt{};
            lt.kind = kLaunchFusedTail;
            lt.attachment = ai;
            lt.aux0 = uint32_t(jobs.size());
            lt.algorithmic_bytes = tiles_at(in_lod) * cc * cc * bpp;
            for (uint32_t k = 1; k <= levels; k++) {
                lt.algorithmic_bytes += tiles_at(in_lod - k) * Tt * Tt * bpp;
                lt.task_count += uint32_t(tiles_at(in_lod - k));
            }
            jobs.push_back(tail);
            plan.push_back(lt);
            in_lod -= levels;
        }

        // cube: aprons that cross a face edge come from the generic stitch kernel (after everything else)
        if (spherical) {
            const uint32_t first = uint32_t(tasks.size());
            for (const Task* t : stitches) {
                const uint32_t n = 1u << t->coord.lod;
                if (t->coord.x != 0 && t->coord.y != 0 && t->coord.x != n - 1 && t->coord.y != n - 1) continue;
                TaskDev d{};
                d.atlas_index = t->atlas_index;
                d.side = t->coord.side;
                d.lod = t->coord.lod;
                d.x = t->coord.x;
                d.y = t->coord.y;
                for (int i = 0; i < 8; i++) {
                    d.rel_index[i] = t->rel[i].atlas_index;
                    d.rel_side[i] = t->rel[i].coordinate.side;
                }
                tasks.push_back(d);
            }
            Launch ls{};
            ls.kind = kLaunchStitch;
            ls.attachment = ai;
            ls.first_task = first;
            ls.task_count = uint32_t(tasks.size()) - first;
            ls.algorithmic_bytes = uint64_t(ls.task_count) * 2 * (2 * m.border_size * (Tt + cc)) * bpp;
            if (ls.task_count) plan.push_back(ls);
        }
    }
    return true;
}

bt_status fused_launch(bt_preprocessor* p, bt_atlas* a, const Launch& l) {
    (void)a;
    FusedJobDev job;
    {
        std::lock_guard<std::mutex> lock(g_jobs_mutex);
        const std::vector<FusedJobDev>& jobs = jobs_of(p);
        if (l.aux0 >= jobs.size()) {
            set_error("fused launch without a plan");
            return BT_ERR_INVALID_ARGUMENT;
        }
        job = jobs[l.aux0];
    }
    job.args.rasters = p->rasters_dev;
    if (l.kind == kLaunchFusedMain) {
        const uint32_t blocks = job.args.item_count * job.args.groups;
        fused_main_kernel<<<blocks, 256, 0, p->ctx->stream>>>(job.args);
    } else {
        const uint32_t size = (1u << job.args.lod) * job.args.m.center_size;
        const dim3 grid((size + 31) / 32, (size + 31) / 32, job.args.sides);
        fused_tail_kernel<<<grid, 256, 0, p->ctx->stream>>>(job.args);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "fused kernel launch");
    return BT_OK;
}

}  // namespace bt
