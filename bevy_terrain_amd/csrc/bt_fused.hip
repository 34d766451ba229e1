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
//   fused_main  : one persistent workgroup per finest-LOD tile (256 threads, a thread = one pair of texture columns), 8 centre rows
//                 per chunk.  While chunk k is shaded out of LDS the source rows of chunk k + 1 travel into the other LDS buffer — by
//                 LDS-DMA (one 1 KiB instruction per source row and wave) where the raster is 16-byte aligned, through registers
//                 otherwise.  Complete, already stitched 1024-byte tile rows of the finest LOD leave as one coalesced store per wave
//                 (aprons are *pulled*: evaluated with the neighbour tile's own formula, hence bit-identical to its centre); row
//                 pairs reduce in registers and lane pairs via DPP to the next two LODs, which are *pushed* into the parent and
//                 grand-parent tiles including the aprons of their x neighbours.  No-data is handled where it is met, per thread
//                 and quad of rows, inside the chunk loop.
//   fused_direct: the same for Rgba8 without LDS staging (a thread = one centre column, software-pipelined 4-row blocks).
//   fused_tail  : the remaining (small) LODs, three per launch, from the atlas: 64 x 64 mosaic pixels per workgroup, 4 x 4 per thread,
//                 no LDS, same push; extra workgroups write the apron rows of the LODs above and, on a cube, the cross-face seam regions
//                 whose sources fused_main produced; the workgroup -> work mapping is XCD-aware.
//   cube seams  : what is left (the top LODs the tail itself produces) comes from stitch_region_kernel afterwards.
//
// Arithmetic contract: identical to bt_kernels.hip / oracle (IEEE binary32, -ffp-contract=off).  The one
// liberty: t / 65535.0f is evaluated as q0 = t*r, e = fma(-q0, 65535, t), q = fma(e, r, q0) with
// r = RN(1/65535) (Markstein's correctly rounded division); equality with `/` for all 65536 inputs is
// checked on the device by bt_selftest() and on the CPU by tests/test_oracle_preprocess.py.
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "bt_internal.hpp"

namespace bt {

namespace {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;

#include "bt_stitch.hpp"  // project_to_side, stitch_source, stitch_region_body (the cube's cross-face seam regions)

// Ablation switches of the profiling build (tools/, -DBT_DEBUG_HOOKS): compiled out of the product library, where
// the branches they guard fold away and no environment variable is read.
#ifdef BT_DEBUG_HOOKS
#define BT_ABLATE(A, bits) ((A).ablate & (bits))
#else
#define BT_ABLATE(A, bits) 0u
#endif
constexpr uint32_t kMainRows = 8;                // centre rows per fused_main workgroup (multiple of 4)
#ifndef BT_DMA_POS
#define BT_DMA_POS 0
#endif

// Where a chunk issues the next chunk's LDS-DMA rows: 0 at its top (the product), 1 / 2 behind its first / second quad of rows, 10 + r behind
// the LDS reads of output row r of the static path.  Round 5, same-lease product builds (profiles/r05_dma_issue_position.txt): behind the
// first quad the clean 16k job is 0 .. 1.8 % faster (bimodal from run to run), behind rows 2 / 3 ~1 % faster with the masked job 2 % slower,
// behind the second quad 3 % slower — inside the noise that matters, so the issue stays at the top.
constexpr uint32_t kDmaPos = BT_DMA_POS;
constexpr uint32_t kMaxBorder = 8;

struct MainItem {  // one finest-LOD tile
    uint32_t side, x, y, atlas_index, raster;
};

struct FusedArgs {
    AttachmentMeta m;
    uint16_t* atlas;
    const RasterDev* rasters;
    const MainItem* items;
    const uint32_t* grids;         // per (side, lod): n x n atlas indices, x-major (x * n + y), INVALID = absent
    uint32_t grid_lod_lo, grid_lod_hi, grid_sides;  // the grids of (side, lod), lod_lo <= lod <= lod_hi, follow each other side-major, LODs ascending
    // A job queued onto a fresh atlas holds its tiles in the reference's allocation order (preprocessor.rs:234-343: per side the finest
    // LOD first, x-major, then each coarser LOD), i.e. atlas index = arithmetic on the coordinate.  regular != 0: the host checked every
    // grid entry against that closed form, and a lookup is a handful of scalar operations instead of a dependent load from the grids
    // (fused_tail was a chain of five dependent round trips, three of them lookups).
    uint32_t regular, reg_first;
    // fused_tail on a cube (round 5): the cross-face apron regions of the LODs fused_main produced — their sources, the neighbour faces'
    // centres, are complete when the tail starts — ride in the tail launch as extra workgroups, one region each (stitch.wgsl:12-51, 79-118),
    // instead of waiting for a launch of their own behind it.  seam_skip: the tail's own apron-row workgroups leave those regions alone.
    const TaskDev* seam_tasks;
    uint32_t seam_count, seam_skip;
    // round 6: the cross-face regions of the LODs the tail ITSELF produces ride in it too — not as copies of the neighbour face's centre (another
    // workgroup of this launch is still writing it) but PULLED: evaluated from the tail's input LOD on the neighbour face with the tail's own
    // reduction (TaskDev::raster = LODs below the input, rel_index[region] = the neighbour tile's x << 16 | y).  seam_pull: the tail's pushes then
    // leave a region beyond exactly one face edge alone instead of clamping into it (a pull workgroup writes it) — and no stitch launch follows.
    uint32_t seam_pull;
    uint32_t tail_extras;  // fused_tail: apron blocks per side (the top / bottom apron rows — Rgba8: and columns — of the LODs above)
    float tlx, tly, brx, bry;
    uint32_t lod;         // finest LOD of this launch (fused_main) / input LOD (fused_tail)
    uint32_t levels;      // LODs produced by this launch: main 1..3 (lod, lod-1, lod-2); tail 1..3 below lod
    uint32_t item_count;  // fused_main: tiles
    uint32_t groups;      // fused_main: row groups per tile
    uint32_t sides;       // fused_tail: 1 or 6
    uint32_t lds_pitch;   // fused_main: texels per staged source row (multiple of 8)
    uint32_t lds_rows;    // fused_main: staged source rows that fit
    uint32_t apron_lods;  // fused_tail: LODs lod, lod+1, ... (this many) get their top / bottom apron rows from extra workgroups
    uint32_t rotate_priority;  // fused_main: wave priority rotates with the chunk index (see fused_main_chunks)
    uint32_t apron_cols;  // fused_tail (Rgba8 after fused_direct, which writes centres only): ... and their left / right apron columns
    // fused_main / fused_direct: every finest tile of the job still holds bt_atlas_create's zeros (Attachment::written, decided by the host per
    // run): "the previous value" of a pixel without data (split.wgsl:34-42) IS 0 — taken as a constant instead of fetched from the atlas.  Both
    // reference examples are this case (clear_attachment, then one dataset per attachment: preprocess_planar.rs:16-60).
    uint32_t prev_zero;
    // fused_main: the tile's 2b apron rows (first / last chunk only) read their source rows from global memory instead of the staged window, which
    // then holds the centre rows alone — set by the host when that is what lets four workgroups share a CU's LDS (source-to-tile ratios from ~1.25)
    uint32_t apron_global;
    // fused_main, run-time-pitch DMA variant: ONE staging buffer — the next chunk's rows are requested when every wave has read this chunk's (no overlap inside the
    // workgroup; the CU's other workgroups cover) — set by the host when two buffers would keep a fourth workgroup off the CU's LDS (ratios from ~1.36 at T = 512)
    uint32_t single_buffer;
    uint32_t ablate;      // debug only (env BT_FUSED_ABLATE): 1 no pyramid, 2 no finest stores, 4 no parent stores, 64 no grand-parent stores (static path), 8 no staging loads, 16 prologue only, 256 / 512 finest / parent stores without arithmetic (use with 16); skeleton shapes: 65536 parent rows in bursts of four chunks, 262144 parent rows as 16-byte stores, 1048576 finest rows as 16-byte stores
};

// t / 65535.0f, correctly rounded, in 3 VALU ops (see header)
__device__ __forceinline__ float unorm16_to_float(uint32_t t) {
    const float x = float(t);
    const float r = 1.0f / 65535.0f;
    const float q0 = x * r;
    const float e = __builtin_fmaf(-q0, 65535.0f, x);
    return __builtin_fmaf(e, r, q0);
}
// floor(0.5 + 65535 * clamp(e, 0, 1)): med3 clamps (no NaN reaches here); the u32 conversion truncates,
// which is floor for the non-negative argument
__device__ __forceinline__ uint32_t float_to_unorm16(float e) {
    const float cl = __builtin_amdgcn_fmed3f(e, 0.0f, 1.0f);
    return uint32_t(0.5f + 65535.0f * cl);
}
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

struct Axis {
    int i0, i1;
    float fr;
};

// split.wgsl:25-32 for centre coordinate r (0..c-1) of tile index `tile` (see bt_kernels.hip split_axis)
// (also evaluated on the host, bit for bit the same IEEE operations, to size the LDS window exactly)
__host__ __device__ __forceinline__ Axis split_axis(uint32_t r, uint32_t c, uint32_t tile, float scale, float lo, float hi, uint32_t dim) {
    const float tc = float(r) / float(c);
    const float s = (float(tile) + tc) / scale;
    const float u = (s - lo) / (hi - lo);
    const float q = u * float(dim) - 0.5f;
    const float fl = floorf(q);
    Axis a;
    a.fr = q - fl;
    const int i = int(fl);
    const int last = int(dim) - 1;
    const int j = i + 1;
    a.i0 = i < 0 ? 0 : (i > last ? last : i);
    a.i1 = j < 0 ? 0 : (j > last ? last : j);
    return a;
}

__device__ __forceinline__ uint32_t grid_lookup(const FusedArgs& A, uint32_t side, uint32_t lod, int x, int y) {
    const int n = int(1u << lod);
    if (x < 0 || y < 0 || x >= n || y >= n) return kInvalid;
    if (lod < A.grid_lod_lo || lod > A.grid_lod_hi || side >= A.grid_sides) return kInvalid;
    if (A.regular) {  // sum of 4^l for lod < l <= lod_hi tiles precede the LOD's on its side, lod_lo .. lod_hi make a side
        const uint32_t hi4 = 4u << (2u * A.grid_lod_hi);
        return A.reg_first + side * ((hi4 - (1u << (2u * A.grid_lod_lo))) / 3u) + (hi4 - (4u << (2u * lod))) / 3u + uint32_t(x) * uint32_t(n) + uint32_t(y);
    }
    // offset of the (side, lod) grid: computed, not looked up (one dependent load less in every lookup chain):
    // sum of 4^l for lod_lo <= l < lod = (4^lod - 4^lod_lo) / 3
    const uint32_t lo4 = 1u << (2u * A.grid_lod_lo), per_side = ((4u << (2u * A.grid_lod_hi)) - lo4) / 3u;
    const uint32_t off = side * per_side + ((1u << (2u * lod)) - lo4) / 3u;
    return A.grids[off + uint32_t(x) * uint32_t(n) + uint32_t(y)];
}

// A tile of some LOD together with its 8 same-face neighbours (stitch.wgsl:57-66 regions N, E, S, W, NW,
// NE, SE, SW).  Out-of-face neighbours count as absent here; on a cube the face-edge tiles are re-stitched
// afterwards by the generic kernel.  Named members: indexed arrays would be demoted to scratch / LDS.
struct TileNb {
    uint32_t self, n, e, s, w;
};

__device__ __forceinline__ TileNb load_tile_nb(const FusedArgs& A, uint32_t side, uint32_t lod, uint32_t x, uint32_t y) {
    const int ix = int(x), iy = int(y);
    TileNb t;
    t.self = grid_lookup(A, side, lod, ix, iy);
    t.n = grid_lookup(A, side, lod, ix, iy - 1);
    t.e = grid_lookup(A, side, lod, ix + 1, iy);
    t.s = grid_lookup(A, side, lod, ix, iy + 1);
    t.w = grid_lookup(A, side, lod, ix - 1, iy);
    return t;
}

// Write centre pixel (cx, cy) of tile (side, lod, tx, ty) — atlas layer `self_index` — and every apron
// texel that copies it (stitch.wgsl:53-118 inverted):
//  - the tile's own apron where the neighbour on that side is absent (repeat_data clamps into the centre),
//  - the facing apron of each existing neighbour whose b-wide strip contains the pixel.
// Neighbours are looked up only for the few pixels within b of a tile edge.
template <bool kCentre = true, typename TT = uint16_t>
__device__ __forceinline__ void push_pixel(const FusedArgs& A, uint32_t side, uint32_t lod, uint32_t tx, uint32_t ty,
                                           uint32_t self_index, uint32_t cx, uint32_t cy, TT v) {
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size;
    const uint64_t tile_texels = uint64_t(T) * T;
    TT* atlas = reinterpret_cast<TT*>(A.atlas);
    TT* self = atlas + uint64_t(self_index) * tile_texels;
    if (kCentre) self[uint64_t(b + cy) * T + b + cx] = v;  // (false: the caller stored the centre texel, e.g. as half of a pair)
    const int ex = cx < b ? -1 : (cx >= c - b ? 1 : 0);
    const int ey = cy < b ? -1 : (cy >= c - b ? 1 : 0);
    if (ex == 0 && ey == 0) return;
    const uint32_t o = b + c;
    const int ix = int(tx), iy = int(ty);
    // a neighbour at offset (dx, dy) sees our centre (cx, cy) at texture (b + cx - dx*c, b + cy - dy*c)
    const uint32_t ax = uint32_t(int(b + cx) - ex * int(c)), ay = uint32_t(int(b + cy) - ey * int(c));
    // (cube, A.seam_pull: a neighbour beyond exactly ONE face edge lives on another face and a pull workgroup of this launch writes that region)
    const int nn = int(1u << lod);
    auto cross_face = [&](int x, int y) { return A.seam_pull != 0 && ((x < 0 || x >= nn) != (y < 0 || y >= nn)); };
    if (ex != 0) {
        const uint32_t n = grid_lookup(A, side, lod, ix + ex, iy);
        if (n != kInvalid) {
            atlas[uint64_t(n) * tile_texels + uint64_t(b + cy) * T + ax] = v;
        } else if (cross_face(ix + ex, iy)) {
        } else if (cx == 0 || cx == c - 1) {  // absent: our outermost column is replicated into our own apron
            const uint32_t x0 = ex < 0 ? 0u : o;
            for (uint32_t j = 0; j < b; j++) self[uint64_t(b + cy) * T + x0 + j] = v;
        }
    }
    if (ey != 0) {
        const uint32_t n = grid_lookup(A, side, lod, ix, iy + ey);
        if (n != kInvalid) {
            atlas[uint64_t(n) * tile_texels + uint64_t(ay) * T + b + cx] = v;
        } else if (cross_face(ix, iy + ey)) {
        } else if (cy == 0 || cy == c - 1) {
            const uint32_t y0 = ey < 0 ? 0u : o;
            for (uint32_t j = 0; j < b; j++) self[uint64_t(y0 + j) * T + b + cx] = v;
        }
    }
    if (ex != 0 && ey != 0) {
        const uint32_t n = grid_lookup(A, side, lod, ix + ex, iy + ey);
        if (n != kInvalid) {
            atlas[uint64_t(n) * tile_texels + uint64_t(ay) * T + ax] = v;
        } else if (cross_face(ix + ex, iy + ey)) {
        } else if ((cx == 0 || cx == c - 1) && (cy == 0 || cy == c - 1)) {  // corner region clamps both axes
            const uint32_t x0 = ex < 0 ? 0u : o, y0 = ey < 0 ? 0u : o;
            for (uint32_t j = 0; j < b; j++)
                for (uint32_t i = 0; i < b; i++) self[uint64_t(y0 + j) * T + x0 + i] = v;
        }
    }
}

// push_pixel in two halves for fused_tail, which is a latency chain: push_targets does the neighbour LOOKUPS of a centre pixel —
// coordinates only, no data — so that they are in flight together with the thread's other loads; push_store does the stores.
// (One memory counter covers loads and stores on this ISA: a lookup issued after a store waits for that store, so four
// push_pixel calls in a row cost an edge thread four round trips.)  A 2 x 2 pixel block with even coordinates shares its
// targets when b is even.
struct PushNb {
    int ex, ey;
    uint32_t nx, ny, nxy;
    uint32_t skip;  // bit 0 / 1 / 2: the x / y / diagonal neighbour lies beyond exactly one face edge of a cube and a pull workgroup writes that region (A.seam_pull)
};
__device__ __forceinline__ PushNb push_targets(const FusedArgs& A, uint32_t side, uint32_t lod, uint32_t tx, uint32_t ty, uint32_t cx, uint32_t cy, bool active) {
    const uint32_t b = A.m.border_size, c = A.m.center_size;
    PushNb t{0, 0, kInvalid, kInvalid, kInvalid, 0u};
    if (!active) return t;
    t.ex = cx < b ? -1 : (cx >= c - b ? 1 : 0);
    t.ey = cy < b ? -1 : (cy >= c - b ? 1 : 0);
    if (t.ex != 0) t.nx = grid_lookup(A, side, lod, int(tx) + t.ex, int(ty));
    if (t.ey != 0) t.ny = grid_lookup(A, side, lod, int(tx), int(ty) + t.ey);
    if (t.ex != 0 && t.ey != 0) t.nxy = grid_lookup(A, side, lod, int(tx) + t.ex, int(ty) + t.ey);
    if (A.seam_pull) {
        const int nn = int(1u << lod), x = int(tx) + t.ex, y = int(ty) + t.ey;
        const bool out_x = x < 0 || x >= nn, out_y = y < 0 || y >= nn;
        t.skip = (t.ex != 0 && out_x ? 1u : 0u) | (t.ey != 0 && out_y ? 2u : 0u) | (t.ex != 0 && t.ey != 0 && out_x != out_y ? 4u : 0u);
    }
    return t;
}
template <typename TT = uint16_t>
__device__ __forceinline__ void push_store(const FusedArgs& A, const PushNb& t, uint32_t self_index, uint32_t cx, uint32_t cy, TT v) {
    if (t.ex == 0 && t.ey == 0) return;
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size, o = b + c;
    const uint64_t tile_texels = uint64_t(T) * T;
    TT* atlas = reinterpret_cast<TT*>(A.atlas);
    TT* self = atlas + uint64_t(self_index) * tile_texels;
    const uint32_t ax = uint32_t(int(b + cx) - t.ex * int(c)), ay = uint32_t(int(b + cy) - t.ey * int(c));
    if (t.ex != 0) {
        if (t.nx != kInvalid) {
            atlas[uint64_t(t.nx) * tile_texels + uint64_t(b + cy) * T + ax] = v;
        } else if (t.skip & 1u) {
        } else if (cx == 0 || cx == c - 1) {
            const uint32_t x0 = t.ex < 0 ? 0u : o;
            for (uint32_t j = 0; j < b; j++) self[uint64_t(b + cy) * T + x0 + j] = v;
        }
    }
    if (t.ey != 0) {
        if (t.ny != kInvalid) {
            atlas[uint64_t(t.ny) * tile_texels + uint64_t(ay) * T + b + cx] = v;
        } else if (t.skip & 2u) {
        } else if (cy == 0 || cy == c - 1) {
            const uint32_t y0 = t.ey < 0 ? 0u : o;
            for (uint32_t j = 0; j < b; j++) self[uint64_t(y0 + j) * T + b + cx] = v;
        }
    }
    if (t.ex != 0 && t.ey != 0) {
        if (t.nxy != kInvalid) {
            atlas[uint64_t(t.nxy) * tile_texels + uint64_t(ay) * T + ax] = v;
        } else if (t.skip & 4u) {
        } else if ((cx == 0 || cx == c - 1) && (cy == 0 || cy == c - 1)) {
            const uint32_t x0 = t.ex < 0 ? 0u : o, y0 = t.ey < 0 ? 0u : o;
            for (uint32_t j = 0; j < b; j++)
                for (uint32_t i = 0; i < b; i++) self[uint64_t(y0 + j) * T + x0 + i] = v;
        }
    }
}

// downsample.wgsl:25-39 on four texels in OFFSETS order (0,0),(0,1),(1,0),(1,1) of (dx, dy)
__device__ __forceinline__ uint32_t downsample4(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11) {
    if (t00 != 0 && t01 != 0 && t10 != 0 && t11 != 0) {
        // all four valid (the common case): ((((0 + a) + b) + c) + d) / 4, and x / 4 == x * 0.25 exactly
        const float value = ((unorm16_to_float(t00) + unorm16_to_float(t01)) + unorm16_to_float(t10)) + unorm16_to_float(t11);
        return float_to_unorm16(value * 0.25f);
    }
    float value = 0.0f, count = 0.0f;
    if (t00 != 0) { value += unorm16_to_float(t00); count += 1.0f; }
    if (t01 != 0) { value += unorm16_to_float(t01); count += 1.0f; }
    if (t10 != 0) { value += unorm16_to_float(t10); count += 1.0f; }
    if (t11 != 0) { value += unorm16_to_float(t11); count += 1.0f; }
    if (count == 0.0f) return 0;
    return float_to_unorm16(value / count);
}

// the same for packed Rgba8 texels: a texel counts when any of r, g, b is non-zero (downsample.wgsl:31), all four
// channels are averaged; operation order of bt_kernels.hip downsample_texel<BT_FORMAT_RGBA8>
__device__ __forceinline__ float unorm8_to_float(uint32_t t) {
    const float x = float(t), r = 1.0f / 255.0f;
    const float q0 = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q0, 255.0f, x), r, q0);
}
__device__ __forceinline__ uint32_t downsample4_rgba8(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11) {
    const uint32_t t[4] = {t00, t01, t10, t11};
    if ((t00 & 0x00FFFFFFu) != 0 && (t01 & 0x00FFFFFFu) != 0 && (t10 & 0x00FFFFFFu) != 0 && (t11 & 0x00FFFFFFu) != 0) {
        // all four count (the common case): ((((0 + a) + b) + c) + d) / 4, and x / 4 == x * 0.25 exactly; the sum of four
        // values in [0, 1] stays in [0, 4], so the clamp of pack4x8unorm is a no-op.  Evaluated in the 2^8-scaled domain:
        // F(t) = fma(x, RN(1 / 255), x) == 256 * RN(t / 255) for all 256 inputs (bt_selftest), additions and the exact
        // factor 0.25 commute with the power-of-two scale, and 255 * v == (255 / 256) * (256 v) as real numbers, so every
        // rounding sees the value the unscaled expression sees — the same texel for a third of the conversion work.
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t sh = 8 * k;
            const float a = float((t00 >> sh) & 0xFFu), bq = float((t01 >> sh) & 0xFFu), cq = float((t10 >> sh) & 0xFFu), d = float((t11 >> sh) & 0xFFu);
            const float r = 1.0f / 255.0f;
            const float sum = ((__builtin_fmaf(a, r, a) + __builtin_fmaf(bq, r, bq)) + __builtin_fmaf(cq, r, cq)) + __builtin_fmaf(d, r, d);
            out |= uint32_t(0.5f + (255.0f / 256.0f) * (sum * 0.25f)) << sh;
        }
        return out;
    }
    float value[4] = {0.0f, 0.0f, 0.0f, 0.0f}, count = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; i++)
        if ((t[i] & 0x00FFFFFFu) != 0) {
#pragma unroll
            for (int k = 0; k < 4; k++) value[k] += unorm8_to_float((t[i] >> (8 * k)) & 0xFFu);
            count += 1.0f;
        }
    if (count == 0.0f) return 0;
    uint32_t out = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float e = value[k] / count;
        const float cl = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
        out |= uint32_t(floorf(0.5f + 255.0f * cl)) << (8 * k);
    }
    return out;
}

// general (slow) evaluation of the finest-LOD mosaic pixel (tile, r) from the source; used for the
// b x b corner aprons only.  `home` = atlas tile holding that pixel (for the keep-previous rule).
__device__ __forceinline__ uint32_t split_value_slow(const FusedArgs& A, const RasterDev& r, uint32_t tx, uint32_t rx, uint32_t ty,
                                                  uint32_t ry, uint32_t home_index) {
    const float scale = float(1u << A.lod);
    const uint32_t c = A.m.center_size, b = A.m.border_size, T = A.m.texture_size;
    const Axis ax = split_axis(rx, c, tx, scale, A.tlx, A.brx, r.width);
    const Axis ay = split_axis(ry, c, ty, scale, A.tly, A.bry, r.height);
    typedef const uint8_t __attribute__((address_space(1))) * global_bytes;
    typedef const uint16_t __attribute__((address_space(1))) * global_u16;
    const global_u16 row0 = (global_u16)((global_bytes)r.data + uint64_t(ay.i0) * r.pitch);
    const global_u16 row1 = (global_u16)((global_bytes)r.data + uint64_t(ay.i1) * r.pitch);
    const uint32_t t00 = row0[ax.i0], t10 = row0[ax.i1], t01 = row1[ax.i0], t11 = row1[ax.i1];
    if (t00 == 0 || t10 == 0 || t01 == 0 || t11 == 0) {
        if (home_index == kInvalid || A.prev_zero) return 0;
        return A.atlas[uint64_t(home_index) * T * T + uint64_t(b + ry) * T + b + rx];
    }
    const float top = mixf(unorm16_to_float(t00), unorm16_to_float(t10), ax.fr);
    const float bot = mixf(unorm16_to_float(t01), unorm16_to_float(t11), ax.fr);
    return float_to_unorm16(mixf(top, bot, ay.fr));
}

// blockIdx -> logical work id such that each XCD (blocks b, b+8, b+16, ... run on XCD b % 8) walks a
// contiguous range of work: neighbouring row groups / tiles then share source halos through one L2.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t total) {
    const uint32_t q = total / 8u, r = total % 8u, xcd = bid % 8u, i = bid / 8u;
    return (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + i;
}

__device__ __forceinline__ int wave_min(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_max(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
    return v;
}

struct RowParam {  // one centre row of the workgroup: LDS slots of its two source rows + the y weight
    int y0, y1;
    float fy;
    uint32_t pad;
};

constexpr uint32_t kMaxChunks = 64;  // chunks one workgroup may run through (T <= 512: a whole tile)
struct MainShared {  // fixed part of the dynamic LDS block (size is a multiple of 16 bytes)
    // row tables of ALL chunks of the workgroup's run, written once in the prologue; index = (k - k_begin) * kMainRows + row
    int row_y0[kMaxChunks * kMainRows];    // first source row; bit 31: the second source row is the same one (clamped)
    float2 row_fy[kMaxChunks * kMainRows];  // y weights (fy, 1 - fy): read by every lane at one address (a broadcast), used as they arrive
    int win_ymin[kMaxChunks];              // source window of a chunk: first row ...
    uint32_t win_slots[kMaxChunks];        // ... and row count; bit 31: the chunk's rows use source rows y, y+1, ..., y+kMainRows
    RowParam apron[2 * kMaxBorder];  // [0, b): top apron rows, [b, 2b): bottom apron rows (pad = mosaic row ry)
    int xmin, xmax;
    uint32_t pad_[2];  // (the dynamic LDS block behind this struct stays 16-byte aligned)
};
static_assert(sizeof(MainShared) % 16 == 0, "LDS carve must stay 16-byte aligned");

struct Texel4 {  // the four source texels a column pair needs from one source row
    float a0, a1, b0, b1;  // converted
    uint32_t za, zb;       // min of the raw pair: 0 <=> a no-data texel in pair a / b
};

__device__ __forceinline__ Texel4 convert4(uint32_t ta0, uint32_t ta1, uint32_t tb0, uint32_t tb1) {
    Texel4 r;
    r.a0 = unorm16_to_float(ta0);
    r.a1 = unorm16_to_float(ta1);
    r.b0 = unorm16_to_float(tb0);
    r.b1 = unorm16_to_float(tb1);
    r.za = min(ta0, ta1);
    r.zb = min(tb0, tb1);
    return r;
}

// the b x b apron corners of the finest tiles: governed by the diagonal neighbour alone (stitch.wgsl:57-66,
// 105-118) — its centre corner if it exists, else the own centre corner — evaluated with the general formula
__device__ __forceinline__ void corner_pixels(const FusedArgs& A, uint32_t item_index, uint32_t tid, uint32_t threads) {
    const MainItem it = A.items[item_index];
    const RasterDev raster = A.rasters[it.raster];
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size, o = b + c;
    const uint32_t self = grid_lookup(A, it.side, A.lod, int(it.x), int(it.y));
    for (uint32_t t = tid; t < 4 * b * b; t += threads) {
        const uint32_t corner = t / (b * b), i = (t % (b * b)) % b, j = (t % (b * b)) / b;
        const bool left = corner == 0 || corner == 3, top = corner < 2;  // 0 NW, 1 NE, 2 SE, 3 SW
        const uint32_t px = left ? i : o + i, py = top ? j : o + j;
        const uint32_t n = grid_lookup(A, it.side, A.lod, int(it.x) + (left ? -1 : 1), int(it.y) + (top ? -1 : 1));
        uint32_t v;
        if (n != kInvalid)
            v = split_value_slow(A, raster, left ? it.x - 1 : it.x + 1, left ? c - b + i : i, top ? it.y - 1 : it.y + 1, top ? c - b + j : j, n);
        else
            v = split_value_slow(A, raster, it.x, left ? 0u : c - 1, it.y, top ? 0u : c - 1, self);
        A.atlas[uint64_t(self) * T * T + py * T + px] = uint16_t(v);
    }
}

// kGeneric == false: the fast variants (packed f32 in the 2^16-scaled domain).  No-data (split.wgsl:34-42, downsample.wgsl:25-39) is detected
// in the texels each thread actually reads, per quad of rows, and fixed in place inside the chunk loop: the thread re-reads the quad's source
// rows from LDS (still staged), derives per-pixel validity, fetches the previous atlas texel for the pixels without data and stores the quad
// as usual; the LOD-1 pixels of such a quad and, where a LOD-1 texel came out 0, the LOD-2 pixel take downsample4's valid-average.  (Rounds
// 1-3 ran the generic rows over flagged chunks as a launch of their own, fused_todo; round 4 redid flagged chunks behind the loop — every one
// staged a second time; both are gone.)  kGeneric == true with kStaged == false reads the source directly (window too large for LDS) and runs
// the generic rows — per-pixel validity for every pixel — inside its loop.
// kDma (fast staged variant, rasters 16-byte aligned — the library pads what it uploads, bt_host.cpp add_raster): the source rows of the next
// chunk travel global -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no commit pass).
template <bool kStaged, bool kGeneric, uint32_t kT, uint32_t kP, bool kDma = false>
__device__ __forceinline__ void fused_main_chunks(const FusedArgs& A, uint32_t item_index, uint32_t k_begin, uint32_t k_end, uint8_t* smem) {
    MainShared& S = *reinterpret_cast<MainShared*>(smem);
    uint16_t* s_buf = reinterpret_cast<uint16_t*>(smem + sizeof(MainShared));
    constexpr bool kFix = kStaged && !kGeneric;  // the fast variants: no-data is detected per thread and quad of rows and fixed in place (round 5)
    const bool one_buffer = kDma && kP == 0 && A.single_buffer != 0;
    const uint32_t buf_texels = one_buffer ? 0u : A.lds_rows * (kP ? kP : A.lds_pitch);  // two staging buffers (chunk parity) — or one: every chunk in the same rows

    // A workgroup is persistent over a run of row chunks (kMainRows centre rows each) of ONE finest tile:
    // the column parameters, neighbour tables and push constants are computed once, and while chunk k is
    // shaded out of LDS the source rows of chunk k + 1 are already in flight into registers.
    const MainItem it = A.items[item_index];
    const RasterDev raster = A.rasters[it.raster];
    // kT / kP: texture size and LDS pitch as compile-time constants (0 = runtime) so that row offsets become
    // instruction immediates in the static fast path
    const uint32_t T = kT ? kT : A.m.texture_size, b = A.m.border_size, c = A.m.center_size, o = b + c;
    const uint32_t tid = threadIdx.x;
#ifdef BT_DEBUG_HOOKS
    // (134217728: real-time (100 MHz) stamps per WORKGROUP — entry, prologue done, chunk loop done, end — behind the per-tile stamps in
    // the atlas's last layer; git history, tools/experiments/main_probe.py)
    auto wg_stamp = [&](uint32_t slot) {
        if (BT_ABLATE(A, 134217728u) && tid == 0 && blockIdx.x < 4096u)
            reinterpret_cast<unsigned long long*>(A.atlas + uint64_t(A.m.atlas_size - 1u) * T * T)[16384u + blockIdx.x * 4u + slot] = __builtin_amdgcn_s_memrealtime();
    };
#else
    auto wg_stamp = [&](uint32_t) {};
#endif
    wg_stamp(0);
    const float scale = float(1u << A.lod);
    const uint32_t tile_texels = T * T;
    const uint32_t chunks_per_tile = (c + kMainRows - 1) / kMainRows;

    const TileNb t5 = load_tile_nb(A, it.side, A.lod, it.x, it.y);
    const uint32_t self4 = A.levels >= 2 ? grid_lookup(A, it.side, A.lod - 1, int(it.x >> 1), int(it.y >> 1)) : kInvalid;
    const uint32_t self3 = A.levels >= 3 ? grid_lookup(A, it.side, A.lod - 2, int(it.x >> 2), int(it.y >> 2)) : kInvalid;

    // row parameters of every chunk of the run, once
    for (uint32_t i = tid; i < (k_end - k_begin) * kMainRows; i += 256u) {
        const uint32_t cr = k_begin * kMainRows + i;
        if (cr < c) {
            const Axis ay = split_axis(cr, c, it.y, scale, A.tly, A.bry, raster.height);
            S.row_y0[i] = ay.i0 | (ay.i1 == ay.i0 ? int(0x80000000u) : 0);
            S.row_fy[i] = float2{ay.fr, 1.0f - ay.fr};
        }
    }
    if (tid >= 32 && tid < 32 + 2 * b) {
        // apron rows: the north / south neighbour's centre rows, or (neighbour absent) clamped into the own centre
        const uint32_t r = tid - 32, k = r % b;
        const bool top = r < b;
        const uint32_t nrow = top ? t5.n : t5.s;
        uint32_t ty, ry;
        if (nrow != kInvalid) { ty = top ? it.y - 1 : it.y + 1; ry = top ? c - b + k : k; } else { ty = it.y; ry = top ? 0u : c - 1; }
        const Axis ay = split_axis(ry, c, ty, scale, A.tly, A.bry, raster.height);
        S.apron[r].y0 = ay.i0;
        S.apron[r].y1 = ay.i1;
        S.apron[r].fy = ay.fr;
        S.apron[r].pad = ry;
    }

    // column pair of this thread: centre pairs first so that lanes (2m, 2m+1) hold the two halves of one
    // pixel of the level two below; the b/2 right and b/2 left apron pairs come last
    const uint32_t half_c = c / 2, half_b = b / 2;
    const bool is_centre = tid < half_c;
    const bool is_right = !is_centre && tid < half_c + half_b;
    const bool is_left = !is_centre && !is_right && tid < half_c + 2 * half_b;
    const bool is_idle = !is_centre && !is_right && !is_left;
    uint32_t px0 = 0;  // texture column of the pair's first pixel
    uint32_t txa = it.x, txb = it.x, rxa = 0, rxb = 0;
    if (is_centre) {
        px0 = b + 2 * tid;
        rxa = 2 * tid;
        rxb = 2 * tid + 1;
    } else if (is_right) {
        const uint32_t j = 2 * (tid - half_c);
        px0 = o + j;
        if (t5.e != kInvalid) { txa = txb = it.x + 1; rxa = j; rxb = j + 1; } else { rxa = rxb = c - 1; }
    } else if (is_left) {
        const uint32_t j = 2 * (tid - half_c - half_b);
        px0 = j;
        if (t5.w != kInvalid) { txa = txb = it.x - 1; rxa = c - b + j; rxb = c - b + j + 1; } else { rxa = rxb = 0; }
    }
    const Axis axa = split_axis(rxa, c, txa, scale, A.tlx, A.brx, raster.width);
    const Axis axb = split_axis(rxb, c, txb, scale, A.tlx, A.brx, raster.width);
    const float fxa = axa.fr, fxb = axb.fr, gxa = 1.0f - fxa, gxb = 1.0f - fxb;
    // atlas tile holding each column's pixels (keep-previous rule reads it when the source has no data)
    const uint32_t home_col = is_right && t5.e != kInvalid ? t5.e : (is_left && t5.w != kInvalid ? t5.w : t5.self);

    // the first left-apron pair reads the leftmost source column, the last right-apron pair the rightmost
    if (tid == half_c + half_b) S.xmin = min(axa.i0, axb.i0);
    if (tid == half_c + half_b - 1) S.xmax = max(axa.i1, axb.i1);
    __syncthreads();
    if (tid < k_end - k_begin) {
        // per chunk: the source window (first row, row count) and whether its rows step through the source one by one
        const uint32_t k = k_begin + tid, n = min(kMainRows, c - k * kMainRows);
        const int* y0 = S.row_y0 + tid * kMainRows;
        const int first = y0[0] & 0x7fffffff;
        bool consecutive = n == kMainRows;
        for (uint32_t i = 0; i < n; i++) consecutive = consecutive && y0[i] == first + int(i);  // also: second row = first + 1
        int lo = first, hi = (y0[n - 1] & 0x7fffffff) + (y0[n - 1] < 0 ? 0 : 1);
        if (!(kDma && kP == 0 && A.apron_global)) {
            if (k == 0) lo = min(lo, S.apron[0].y0);
            if (k == chunks_per_tile - 1) hi = max(hi, S.apron[2 * b - 1].y1);
        }
        S.win_ymin[tid] = lo;
        S.win_slots[tid] = uint32_t(hi - lo + 1) | (consecutive ? 0x80000000u : 0u);
    }
    __syncthreads();

    // source window of a chunk: columns [xa, xa + pitch) with xa 8-texel aligned (the same for every chunk);
    // LDS slot of source row y = y - ymin(chunk) (the rows of a chunk are consecutive mosaic rows; the host
    // sized the LDS for their contiguous range)
    const int xa = __builtin_amdgcn_readfirstlane(S.xmin) & ~7;
    const uint32_t P = kP ? kP : A.lds_pitch;
    typedef const uint8_t __attribute__((address_space(1))) * global_bytes;
    typedef const uint16_t __attribute__((address_space(1))) * global_u16;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef const u32x4 __attribute__((address_space(1))) * global_u4;
    typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
    const global_bytes data = (global_bytes)raster.data;  // loaded from memory: tell the compiler it is global
    const uint32_t chunks_per_row = P / 8u;
    const uint32_t row_texels = uint32_t(raster.pitch / 2u);  // addressable texels per source row
    // (16-byte pieces: base and pitch 16-byte aligned — a row then ends on a piece boundary and no piece straddles it.  The library pads the
    // rasters it uploads and copies a borrowed unaligned device raster once, bt_host.cpp add_raster: the texel-by-texel staging below is left
    // for deferred host rasters of odd width)
    const bool wide = ((reinterpret_cast<uintptr_t>(raster.data) | raster.pitch) & 15u) == 0;
    constexpr uint32_t kBatch = 4;  // 16-byte loads per thread and chunk (host guarantees slots * pitch / 8 <= 256 * kBatch)

    auto chunk_rows = [&](uint32_t k) -> uint32_t { return min(kMainRows, c - k * kMainRows); };
    auto window = [&](uint32_t k, int& ymin, uint32_t& slots) {  // workgroup-uniform
        ymin = __builtin_amdgcn_readfirstlane(S.win_ymin[k - k_begin]);
        slots = uint32_t(__builtin_amdgcn_readfirstlane(S.win_slots[k - k_begin])) & 0x7fffffffu;
    };
    // This thread's kBatch 16-byte pieces of a staged window never change: piece i is column group kk_i of window
    // row slot_i.  Source byte offset from the window's first row and LDS byte offset, computed once per tile;
    // a chunk then adds only its uniform base address (saddr + 32-bit voffset loads, no per-chunk address math).
    uint32_t src_off[kBatch], lds_off[kBatch];
#pragma unroll
    for (uint32_t i = 0; i < kBatch; i++) {
        const uint32_t ch = tid + 256u * i;
        const uint32_t slot = ch / chunks_per_row, kk = ch - slot * chunks_per_row;
        const uint32_t x = min(uint32_t(xa) + 8u * kk, row_texels - 8u);  // x, row_texels: multiples of 8
        src_off[i] = slot * uint32_t(raster.pitch) + x * 2u;  // host: lds_rows * pitch < 2^31
        lds_off[i] = (slot * P + 8u * kk) * 2u;
    }
    const uint32_t safe_off = src_off[0] - (tid / chunks_per_row) * uint32_t(raster.pitch);  // same columns, window row 0
    // issue the loads of a chunk into registers (branch-free: pieces past the window re-read a piece of its first row)
    auto stage_issue = [&](int ymin, uint32_t slots, u32x4 (&v)[kBatch]) {
        const global_bytes base = data + uint64_t(uint32_t(ymin)) * raster.pitch;  // uniform
        const uint32_t bound = slots * P * 2u;
#pragma unroll
        for (uint32_t i = 0; i < kBatch; i++) {
            const global_u4 ptr = (global_u4)(base + (lds_off[i] < bound ? src_off[i] : safe_off));
            v[i] = BT_ABLATE(A, 32768u) ? __builtin_nontemporal_load(ptr) : *ptr;  // (32768: streaming loads, timing experiment)
        }
    };
    // registers -> LDS
    auto stage_commit = [&](uint16_t* s_src, uint32_t slots, const u32x4 (&v)[kBatch]) {
        const uint32_t bound = slots * P * 2u;
        uint8_t* s_bytes = reinterpret_cast<uint8_t*>(s_src);
#pragma unroll
        for (uint32_t i = 0; i < kBatch; i++)
            if (lds_off[i] < bound) *reinterpret_cast<u32x4*>(s_bytes + lds_off[i]) = v[i];
    };
    // unaligned rasters (odd widths / pitches): texel by texel, no prefetch
    auto stage_narrow = [&](uint16_t* s_src, int ymin, uint32_t slots) {
        for (uint32_t i = tid; i < slots * P; i += 256u) {
            const uint32_t slot = i / P, kk = i - slot * P;
            const uint32_t x = uint32_t(xa) + kk;
            const global_u16 row = (global_u16)(data + uint64_t(ymin + int(slot)) * raster.pitch);
            s_src[i] = x < raster.width ? row[x] : uint16_t(1);
        }
    };

    // LDS-DMA staging of a window: wave w moves window rows w, w + 4, ...; a row is one 1 KB instruction (64 lanes x 16
    // bytes, LDS destination = uniform base + lane * 16) plus a 2-lane instruction for the last 32 bytes of the 1056-byte
    // row; pieces past the raster row's end re-read its last piece like the register path (never used: the column
    // indices are clamped into the raster)
    typedef __attribute__((address_space(3))) uint8_t* lds_bytes;
    const uint32_t dma_lane = tid & 63u;
    const uint32_t dma_off_main = min(uint32_t(xa) * 2u + dma_lane * 16u, uint32_t(raster.pitch) - 16u);
    const uint32_t dma_off_tail = min(uint32_t(xa) * 2u + 1024u + (dma_lane & 1u) * 16u, uint32_t(raster.pitch) - 16u);
    auto dma_issue = [&](uint16_t* s_dst, int ymin, uint32_t slots) {
        static_assert(!kDma || kP == 528 || kP == 0, "the DMA variants: 1056-byte LDS rows, or a run-time pitch");
        const lds_bytes dst = (lds_bytes)reinterpret_cast<uint8_t*>(s_dst);
        // (the wave index through v_readfirstlane: as a function of tid the row pointer was computed per lane — two 64-bit vector
        // multiply-adds per row; every instruction, scalar or vector, takes a turn of the SIMD's one issue port: git history, tools/experiments/issue_probe.hip)
        for (uint32_t slot = uint32_t(__builtin_amdgcn_readfirstlane(int(tid >> 6))); slot < slots; slot += 4u) {
            const global_bytes row = data + uint64_t(uint32_t(ymin) + slot) * raster.pitch;
            if constexpr (kP == 528) {
                if (BT_ABLATE(A, 32768u))  // (32768: the non-temporal policy on the source stream — timing experiment)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + dma_off_main),
                                                     (__attribute__((address_space(3))) void*)(dst + slot * (kP * 2u)), 16, 0, 2);
                else
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + dma_off_main),
                                                 (__attribute__((address_space(3))) void*)(dst + slot * (kP * 2u)), 16, 0, 0);
                if (dma_lane < 2u)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + dma_off_tail),
                                                     (__attribute__((address_space(3))) void*)(dst + slot * (kP * 2u) + 1024u), 16, 0, 0);
            } else {
                // a run-time pitch (a source-to-tile ratio away from 1: wider and more rows than the register staging can batch): the row in
                // 1 KB pieces, the last one with the lanes it has texels for
                const uint32_t row_bytes = P * 2u;  // a multiple of 16
                for (uint32_t piece = 0; piece < row_bytes; piece += 1024u)
                    if (piece + dma_lane * 16u < row_bytes)
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(row + min(uint32_t(xa) * 2u + piece + dma_lane * 16u, uint32_t(raster.pitch) - 16u)),
                                                         (__attribute__((address_space(3))) void*)(dst + slot * row_bytes + piece), 16, 0, 0);
            }
        }
    };

    uint16_t* tile5 = A.atlas + uint64_t(t5.self) * tile_texels;
    uint32_t* tile5_u32 = reinterpret_cast<uint32_t*>(tile5);
    uint16_t* tile4 = A.atlas + uint64_t(self4 == kInvalid ? 0u : self4) * tile_texels;
    uint16_t* tile3 = A.atlas + uint64_t(self3 == kInvalid ? 0u : self3) * tile_texels;
    const uint32_t cx4 = (it.x & 1u) * half_c + tid, cy4_base = (it.y & 1u) * half_c;
    const uint32_t cx3 = (it.x & 3u) * (c / 4) + (tid >> 1), cy3_base = (it.y & 3u) * (c / 4);
    const bool do4 = A.levels >= 2 && !BT_ABLATE(A, 1u), do3 = A.levels >= 3 && !BT_ABLATE(A, 1u);
    // Left / right apron columns of the parent (shift 1) and grand-parent (shift 2) tile: a centre column within b of
    // the tile's x edge is also the x neighbour's apron column (stitch.wgsl:79-88); with that neighbour absent the
    // own apron repeats the edge column (stitch.wgsl:105-118) and the edge column's thread writes all b of them.
    // Per thread and level: element offset of the first extra texel in the row of centre row 0, and how many.
    // (the target layer and the offset inside it travel separately: layer x tile texels passes 2^32 in an atlas of more than 16384 tiles of 512^2)
    auto make_xpush = [&](uint32_t shift, uint32_t self, uint32_t cx, uint32_t& layer, uint32_t& off, uint32_t& count) {
        layer = 0;
        off = 0;
        count = 0;
        if (!is_centre || self == kInvalid) return;
        // cx is the column in the PARENT tile (it already carries this finest tile's share of it): with narrow tiles
        // (c / 2^shift < b) the strip of width b spans more than one finest tile's share
        const bool left = cx < b, right = cx >= c - b;
        if (!left && !right) return;
        const uint32_t n = grid_lookup(A, it.side, A.lod - shift, int(it.x >> shift) + (left ? -1 : 1), int(it.y >> shift));
        const uint32_t j = left ? cx : cx - (c - b);
        if (n != kInvalid) {
            layer = n;
            off = b * T + (left ? o + j : j);
            count = 1;
        } else if (left ? cx == 0 : cx == c - 1) {
            layer = self;
            off = b * T + (left ? 0u : o);
            count = b;
        }
    };
    uint32_t x4_layer, x4_off, x4_count, x3_layer, x3_off, x3_count;
    make_xpush(1, A.levels >= 2 ? self4 : kInvalid, cx4, x4_layer, x4_off, x4_count);
    make_xpush(2, A.levels >= 3 ? self3 : kInvalid, cx3, x3_layer, x3_off, x3_count);
    auto xpush4 = [&](uint32_t cy, uint16_t v) {  // cy: centre row in the parent tile
        for (uint32_t e = 0; e < x4_count; e++) A.atlas[uint64_t(x4_layer) * tile_texels + x4_off + cy * T + e] = v;
    };
    auto xpush3 = [&](uint32_t cy, uint16_t v) {
        for (uint32_t e = 0; e < x3_count; e++) A.atlas[uint64_t(x3_layer) * tile_texels + x3_off + cy * T + e] = v;
    };
    // LDS offsets of this thread's four source columns (idle lanes read column 0 and store nothing)
    const uint32_t la0 = uint32_t(axa.i0 - xa), la1 = uint32_t(axa.i1 - xa), lb0 = uint32_t(axb.i0 - xa), lb1 = uint32_t(axb.i1 - xa);

    // ---- first chunk: stage synchronously
    int ymin = 0;
    uint32_t slots = 0;
    u32x4 pre[kBatch];
    window(k_begin, ymin, slots);
    if constexpr (kDma) {
        dma_issue(s_buf + (k_begin & 1u) * buf_texels, ymin, slots);
    } else if (kStaged && !BT_ABLATE(A, 8u)) {
        if (wide) {
            stage_issue(ymin, slots, pre);
            stage_commit(s_buf + (k_begin & 1u) * buf_texels, slots, pre);
        } else {
            stage_narrow(s_buf + (k_begin & 1u) * buf_texels, ymin, slots);
        }
    }
    // the tile's b x b apron corners (4 b^2 pixels by the general formula): evaluated while the first rows are on their way
    if constexpr (kStaged) {
        if (k_begin == 0) corner_pixels(A, item_index, tid, 256u);
    }
    // the chunk barrier: the staged rows of the next chunk are visible behind it (kDma: behind the landing of this wave's DMA rows —
    // and, one memory counter, of its stores).  No no-data flag travels any more: every fast variant fixes a no-data quad where it meets it.
    auto chunk_barrier = [&]() {
        if constexpr (kDma) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    // ---- per-chunk pieces
    uint16_t* s_src = s_buf;  // staged rows of the chunk being shaded ...
    int cur_ymin = 0;         // ... and the source row of its first slot
    auto fetch_row = [&](int y) -> Texel4 {
        if constexpr (kStaged) {
            const uint16_t* row = s_src + uint32_t(y - cur_ymin) * P;
            return convert4(row[la0], row[la1], row[lb0], row[lb1]);
        } else {
            const global_u16 row = (global_u16)(data + uint64_t(y) * raster.pitch);
            return convert4(row[axa.i0], row[axa.i1], row[axb.i0], row[axb.i1]);
        }
    };
    // apron rows (first / last chunk of a tile only): the centre columns are rows like any other (the north / south
    // neighbour's centre rows, or clamped into the own centre); the b x b corners follow the diagonal neighbour alone
    // (stitch.wgsl:57-66, 105-118) and are written by corner_pixels.  gtag: with the keep-previous rule (generic rows) or without
    auto apron_rows = [&](uint32_t k, auto gtag) {
        constexpr bool kKeep = decltype(gtag)::value;
        if (!((k == 0 || k == chunks_per_tile - 1) && is_centre)) return;
        for (uint32_t r = 0; r < 2 * b; r++) {
            const bool top = r < b;
            if (top ? k != 0 : k != chunks_per_tile - 1) continue;
            const uint32_t kk = r % b, py = top ? kk : o + kk;
            const int y0 = __builtin_amdgcn_readfirstlane(S.apron[r].y0), y1 = __builtin_amdgcn_readfirstlane(S.apron[r].y1);
            const float fy = S.apron[r].fy, gy = 1.0f - fy;
            Texel4 t0, t1;
            if (kDma && kP == 0 && A.apron_global) {  // (the run-time-pitch DMA variant only; the window holds the centre rows only: see FusedArgs::apron_global)
                const global_u16 r0 = (global_u16)(data + uint64_t(y0) * raster.pitch), r1 = (global_u16)(data + uint64_t(y1) * raster.pitch);
                t0 = convert4(r0[axa.i0], r0[axa.i1], r0[axb.i0], r0[axb.i1]);
                t1 = convert4(r1[axa.i0], r1[axa.i1], r1[axb.i0], r1[axb.i1]);
            } else {
                t0 = fetch_row(y0);
                t1 = fetch_row(y1);
            }
            uint32_t va = float_to_unorm16((t0.a0 * gxa + t0.a1 * fxa) * gy + (t1.a0 * gxa + t1.a1 * fxa) * fy);
            uint32_t vb = float_to_unorm16((t0.b0 * gxb + t0.b1 * fxb) * gy + (t1.b0 * gxb + t1.b1 * fxb) * fy);
            if (kKeep && min(min(t0.za, t1.za), min(t0.zb, t1.zb)) == 0) {
                const uint32_t nrow = top ? t5.n : t5.s;
                const uint16_t* h = A.atlas + uint64_t(nrow == kInvalid ? t5.self : nrow) * tile_texels + (b + S.apron[r].pad) * T + b;
                if (A.prev_zero) {
                    if (min(t0.za, t1.za) == 0) va = 0;
                    if (min(t0.zb, t1.zb) == 0) vb = 0;
                } else {
                    if (min(t0.za, t1.za) == 0) va = h[rxa];
                    if (min(t0.zb, t1.zb) == 0) vb = h[rxb];
                }
                asm volatile("" : "+v"(va), "+v"(vb));  // (the values arrive inside the rare branch: no wait for everything in flight behind it)
            }
            tile5_u32[(py * T + px0) >> 1] = va | (vb << 16);
        }
    };
    // the generic rows of chunk k: per-pixel validity (no-data texels), keep-previous rule, valid-average
    auto generic_rows = [&](uint32_t k) {
        const int* row_y0 = S.row_y0 + (k - k_begin) * kMainRows;
        const float2* row_fy = S.row_fy + (k - k_begin) * kMainRows;
        const uint32_t cr0 = k * kMainRows, nrows = chunk_rows(k);
        // ---- generic loop: tracks per-pixel validity (no-data texels), keep-previous rule, valid-average
        Texel4 cur{};
        int cur_y = -1;
        for (uint32_t q = 0; q < nrows; q += 4) {
            uint32_t va[4], vb[4];
            uint32_t zany = 1;
            // (only the unstaged variant calls this: the staged ones fix a no-data quad inside their own loops since round 5)
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) {
                const int yy = __builtin_amdgcn_readfirstlane(row_y0[q + i]);
                const int y0 = yy & 0x7fffffff, y1 = y0 + (yy < 0 ? 0 : 1);
                const float fy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, row_fy[q + i].x)));
                const Texel4 top = (y0 == cur_y) ? cur : fetch_row(y0);
                const Texel4 bot = (y1 == y0) ? top : fetch_row(y1);
                cur = bot;
                cur_y = y1;
                const float gy = 1.0f - fy;
                va[i] = float_to_unorm16((top.a0 * gxa + top.a1 * fxa) * gy + (bot.a0 * gxa + bot.a1 * fxa) * fy);
                vb[i] = float_to_unorm16((top.b0 * gxb + top.b1 * fxb) * gy + (bot.b0 * gxb + bot.b1 * fxb) * fy);
                const uint32_t za = min(top.za, bot.za), zb = min(top.zb, bot.zb);
                zany = min(zany, min(za, zb));
                // remember the validity in bit 16 (cleared below): 0x10000 = no data in the footprint
                va[i] |= za == 0 ? 0x10000u : 0u;
                vb[i] |= zb == 0 ? 0x10000u : 0u;
            }
            if (zany == 0) {  // some footprint had no data: those pixels keep their previous atlas value (split.wgsl:37-42)
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint16_t* h = A.atlas + uint64_t(home_col) * tile_texels + (b + cr0 + q + i) * T + b;
                    if (va[i] & 0x10000u) va[i] = A.prev_zero ? 0u : uint32_t(h[rxa]);
                    if (vb[i] & 0x10000u) vb[i] = A.prev_zero ? 0u : uint32_t(h[rxb]);
                }
            }
            const uint32_t py = b + cr0 + q;
            if (!is_idle) {
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) tile5_u32[((py + i) * T + px0) >> 1] = (va[i] & 0xFFFFu) | (vb[i] << 16);
            }
            if (do4) {
                const uint32_t cy = cr0 + q;  // multiple of 4
                const uint32_t q0 = downsample4(va[0] & 0xFFFFu, va[1] & 0xFFFFu, vb[0] & 0xFFFFu, vb[1] & 0xFFFFu);  // OFFSETS order
                const uint32_t q1 = downsample4(va[2] & 0xFFFFu, va[3] & 0xFFFFu, vb[2] & 0xFFFFu, vb[3] & 0xFFFFu);
                const uint32_t cy4 = cy4_base + (cy >> 1);
                if (is_centre) {
                    uint16_t* dst = tile4 + (b + cy4) * T + b + cx4;
                    dst[0] = uint16_t(q0);
                    dst[T] = uint16_t(q1);
                }
                if (x4_count) {
                    xpush4(cy4, uint16_t(q0));
                    xpush4(cy4 + 1, uint16_t(q1));
                }
                if (do3) {
                    const uint32_t other0 = __shfl_xor(q0, 1), other1 = __shfl_xor(q1, 1);
                    if (is_centre && (tid & 1u) == 0) {
                        const uint32_t w = downsample4(q0, q1, other0, other1);
                        tile3[(b + cy3_base + (cy >> 2)) * T + b + cx3] = uint16_t(w);
                        if (x3_count) xpush3(cy3_base + (cy >> 2), uint16_t(w));
                    }
                }
            }
        }
    };

    chunk_barrier();
    wg_stamp(1);

    for (uint32_t k = k_begin; k < k_end; k++) {
        const int* row_y0 = S.row_y0 + (k - k_begin) * kMainRows;
        const float2* row_fy = S.row_fy + (k - k_begin) * kMainRows;
        const uint32_t cr0 = k * kMainRows, nrows = chunk_rows(k);
        s_src = s_buf + (k & 1u) * buf_texels;  // this chunk's staged rows; the other half receives chunk k + 1
        cur_ymin = ymin;
        // prefetch the next chunk's source rows while this one is shaded
        int next_ymin = 0;
        uint32_t next_slots = 0;
        const bool more = k + 1 < k_end;
        if (more) {
            window(k + 1, next_ymin, next_slots);
            if constexpr (kDma) {
                if (BT_ABLATE(A, 2048u)) __builtin_amdgcn_s_setprio(3);  // (2048: the DMA issue at top priority — timing experiment)
                if (kDmaPos == 0 && !one_buffer) dma_issue(s_buf + ((k + 1) & 1u) * buf_texels, next_ymin, next_slots);
            }
            else if (kStaged && wide && !BT_ABLATE(A, 8u)) stage_issue(next_ymin, next_slots, pre);
        }

        // Wave priority rotating with the chunk index, offset by the workgroup's dispatch rank on its CU.  The CU's arbiters serve
        // the highest-priority wave first and, among equals, the OLDEST — strictly: without this the four resident workgroups
        // of a CU finish their tiles after 190 / 215 / 243 / 272 us in dispatch order (git history, tools/experiments/drift_probe.py; static priorities by
        // rank reverse the staircase exactly).  Giving every workgroup a quarter of its chunks at each level makes them finish
        // within 10 us of each other and the kernel 2.3 % shorter (283.4 -> 276.9 us).  Two jobs in flight on two contexts lose
        // ~3 % with it (the staircase lets the next job's workgroups move in early): the single job is what is optimised.
        if (A.rotate_priority) {
            switch (((blockIdx.x / 8u) / 32u + k) & 3u) {
                case 0: __builtin_amdgcn_s_setprio(0); break;
                case 1: __builtin_amdgcn_s_setprio(1); break;
                case 2: __builtin_amdgcn_s_setprio(2); break;
                default: __builtin_amdgcn_s_setprio(3); break;
            }
        }
#ifdef BT_DEBUG_HOOKS
#define BT_FUSED_DEBUG_CHUNK_PROBES
#include "bt_fused_debug.inc"  // (per-chunk time stamps, rotating wave priorities: timing experiments)
#undef BT_FUSED_DEBUG_CHUNK_PROBES
#endif
        apron_rows(k, std::integral_constant<bool, true>{});  // (every variant handles no-data where it is met: the keep-previous form)

#ifdef BT_DEBUG_HOOKS
#define BT_FUSED_DEBUG_SKELETON_STORES
#include "bt_fused_debug.inc"  // (memory-skeleton store shapes of the profiling build)
#undef BT_FUSED_DEBUG_SKELETON_STORES
#endif
        if (!BT_ABLATE(A, 16u)) {
            if constexpr (kStaged && !kGeneric) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                const f2 gx = {gxa, gxb}, fx = {fxa, fxb};
                // The fast loop works on texel values scaled by 2^16: F(t) = 65536 * RN(t / 65535).  Every later
                // operation (weighted sums, x 0.25) is homogeneous in the texel values and a power-of-two scale commutes
                // with IEEE rounding (no overflow / underflow here: values in [1, 65536], weights in [0, 1]), so every
                // intermediate is exactly 65536 x the oracle's, and 65535 * v == (65535 / 65536) * V as real numbers with
                // 65535 / 65536 exactly representable: the quantised results are bit-identical.  What it buys: F(t) is ONE
                // operation, fma(x, r, x) with x = float(t), r = RN(1 / 65535) — a single rounding of x + x * r, equal
                // to 65536 * RN(t / 65535) for all 65536 inputs (checked on the device by bt_selftest and exactly, in
                // rational arithmetic, by tests/test_oracle_preprocess.py) — instead of the three of the unscaled division.
                const f2 kr = {1.0f / 65535.0f, 1.0f / 65535.0f}, kn = {65535.0f / 65536.0f, 65535.0f / 65536.0f}, khalf = {0.5f, 0.5f};
                const f2 kzero = {0.0f, 0.0f}, knq = {0.25f * (65535.0f / 65536.0f), 0.25f * (65535.0f / 65536.0f)};
                auto conv2 = [&](uint32_t ta, uint32_t tb) -> f2 {  // (F(ta), F(tb))
                    const f2 x = {float(ta), float(tb)};
                    return __builtin_elementwise_fma(x, kr, x);
                };
                // 0.5 + 65535 * clamp(v, 0, 1) with v = V / 65536; the u32 conversion then floors.  In this loop every
                // input lies in (0, 1] and the weights in [0, 1], so v is in (0, 1 + a few ulp]: the clamp can only act on an
                // excess of ~1e-7, and floor(0.5 + 65535 * (1 + 1e-7)) = 65535 = floor(0.5 + 65535 * 1) — it is a no-op on
                // the result and is left out (tests: saturated rasters, all-65535 blocks).
                auto quantise = [&](f2 v) -> f2 { return khalf + kn * v; };
                // the same for a sum of four: (s * 0.25) is exact, so RN((s * 0.25) * k) == RN(s * (0.25 * k))
                auto quantise_quarter = [&](f2 s) -> f2 { return khalf + knq * s; };
                const uint32_t cy4_first = cy4_base + (cr0 >> 1), cy3_first = cy3_base + (cr0 >> 2);
                if (kMainRows == 8 && __builtin_amdgcn_readfirstlane(int(S.win_slots[k - k_begin])) < 0) {
                    // ---- static fast path: the 8 rows use 9 consecutive source rows: straight-line code, row offsets
                    // are immediates, no per-row decisions.
                    // Rolling over the source rows (two live horizontal blends) in two quads of output rows; the
                    // four texels of the next source row are requested before the current row is converted, so the
                    // LDS latency hides behind the packed arithmetic.
                    const uint16_t* base = s_src + uint32_t(__builtin_amdgcn_readfirstlane(row_y0[0]) - cur_ymin) * P;  // consecutive: bit 31 clear
                    const uint16_t *pa0 = base + la0, *pa1 = base + la1, *pb0 = base + lb0, *pb1 = base + lb1;
                    const float2* fyt = row_fy;
                    // the texels of a source row as two 16-bit PAIRS (columns a | b at i0, at i1): the pairs convert with SDWA selects
                    // and the no-data test is two packed 16-bit minima per row
                    u16x2 ta = {pa0[0], pb0[0]}, tb = {pa1[0], pb1[0]};
                    u16x2 na = {pa0[P], pb0[P]}, nb = {pa1[P], pb1[P]};
                    auto conv2p = [&](u16x2 t) -> f2 { return conv2(t.x, t.y); };
                    f2 hprev = conv2p(ta) * gx + conv2p(tb) * fx;
                    uint32_t q[4];
                    // smallest raw texel this thread read for the rows of a quad (source rows 4 * quad .. 4 * quad + 4), per column pair half
                    u16x2 zmin[2] = {__builtin_elementwise_min(ta, tb), u16x2{0xFFFFu, 0xFFFFu}};
                    // ... and per source row of the current quad (z[j]: source row 4 * quad + j): computed anyway, kept in registers until the quad's test so
                    // that a no-data quad derives its per-pixel validity without reading the rows from LDS again (round 6: 20 LDS reads per such quad)
                    u16x2 z[5] = {zmin[0], zmin[0], zmin[0], zmin[0], zmin[0]};
                    uint32_t zq[2] = {1u, 1u};
                    uint32_t sink[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};  // (ablation 1073741824 only)
                    uint32_t* dst5 = tile5_u32 + (((b + cr0) * T + px0) >> 1);
#pragma unroll
                    for (uint32_t quad = 0; quad < 2; quad++) {
                        uint32_t ua[4], ub[4];
#pragma unroll
                        for (uint32_t i = 0; i < 4; i++) {
                            const uint32_t r = 4 * quad + i;  // output row; needs source rows r and r + 1
                            ta = na;
                            tb = nb;
                            if (r + 2 <= kMainRows) {
                                na = u16x2{pa0[(r + 2) * P], pb0[(r + 2) * P]};
                                nb = u16x2{pa1[(r + 2) * P], pb1[(r + 2) * P]};
                            }
                            if (kDma && kDmaPos == 10 + r && more) dma_issue(s_buf + ((k + 1) & 1u) * buf_texels, next_ymin, next_slots);
                            if constexpr (kFix) {
                                const u16x2 zrow = __builtin_elementwise_min(ta, tb);
                                zmin[quad] = __builtin_elementwise_min(zmin[quad], zrow);
                                if (quad == 0 && i == 3) zmin[1] = zrow;  // source row 4 feeds both quads
                                if (quad == 1 && i == 0) z[0] = z[4];     // (its minimum is the second quad's first row)
                                z[i + 1] = zrow;
                            }
                            const f2 hnew = conv2p(ta) * gx + conv2p(tb) * fx;
                            // (fy, 1 - fy) of output row r: one 8-byte LDS read at a workgroup-uniform address, the values stay in
                            // vector registers (no v_readfirstlane, no subtraction: 16 VALU instructions less per chunk)
                            // (a float2 struct copy on purpose.  It carries no type tag, so the compiler cannot tell this read from the LDS-DMA rows in
                            // flight into the other staging buffer and puts s_waitcnt vmcnt(0) in front of the first one: every chunk waits for the rows
                            // it has just requested before it shades.  Read as two typed floats the wait is gone — same instructions otherwise — and
                            // the 16k job is 0.8 % SLOWER, same lease, 3 + 3 runs: the other waves of the CU cover the wait, and waves that start a
                            // chunk together keep the row stores of a workgroup together.  profiles/r06_static_wait.txt)
                            const float2 wy = fyt[r];
                            const f2 fy2 = {wy.x, wy.x}, gy2 = {wy.y, wy.y};
                            const f2 w = quantise(hprev * gy2 + hnew * fy2);
                            hprev = hnew;
                            ua[i] = uint32_t(w.x);
                            ub[i] = uint32_t(w.y);
                        }
                        if constexpr (kFix) {
                            zq[quad] = min(uint32_t(zmin[quad].x), uint32_t(zmin[quad].y));
                            if (__builtin_expect(zq[quad] == 0, 0)) {
                                // (rare) this thread read a no-data texel for the quad: per-pixel validity from the rows' minima (kept above), the
                                // previous atlas value where a footprint has no data (split.wgsl:34-42) — in place, no redo
                                const uint16_t* h = A.atlas + uint64_t(home_col) * tile_texels + (b + cr0 + 4 * quad) * T + b;
                                if (A.prev_zero) {  // a fresh atlas: the previous value is bt_atlas_create's 0 — no fetch, no wait
#pragma unroll
                                    for (uint32_t i = 0; i < 4; i++) {
                                        if (min(z[i].x, z[i + 1].x) == 0) ua[i] = 0;
                                        if (min(z[i].y, z[i + 1].y) == 0) ub[i] = 0;
                                    }
                                } else if (BT_ABLATE(A, 1073741824u)) {
                                    // (1073741824, timing only, WRONG tiles: the fetches are issued but nobody waits for them before the chunk's
                                    // barrier — the bound of any scheme that defers a no-data quad: profiles/r06_masked16k.txt)
#pragma unroll
                                    for (uint32_t i = 0; i < 4; i++) {
                                        if (min(z[i].x, z[i + 1].x) == 0) ua[i] = 0;
                                        if (min(z[i].y, z[i + 1].y) == 0) ub[i] = 0;
                                        asm volatile("global_load_ushort %0, %1, off" : "=v"(sink[2 * i]) : "v"(h + i * T + rxa) : "memory");
                                        asm volatile("global_load_ushort %0, %1, off" : "=v"(sink[2 * i + 1]) : "v"(h + i * T + rxb) : "memory");
                                    }
                                } else {
                                    // all eight previous values of the quad, unconditionally and back to back, THEN the selects: a conditional load is a
                                    // divergent block of its own with a wait behind it — up to eight dependent round trips per no-data quad (round 6)
                                    // (conditional loads on purpose.  Loading the quad's eight previous values unconditionally and back to back — one wait
                                    // instead of up to eight — takes 2.5 % off the re-run of the masked 16k job, but the kernel then allocates 125 VGPRs instead
                                    // of 114 and the CLEAN job comes out 0.5 - 1.6 % slower, same-lease, as 8 x u16, 4 + 4 and 4 x dword loads alike: round 6)
#pragma unroll
                                    for (uint32_t i = 0; i < 4; i++) {
                                        if (min(z[i].x, z[i + 1].x) == 0) ua[i] = h[i * T + rxa];
                                        if (min(z[i].y, z[i + 1].y) == 0) ub[i] = h[i * T + rxb];
                                    }
                                }
                                // the fetched values arrive HERE, inside the rare branch: left pending, the compiler guards the stores behind the
                                // branch — on the fast path too — with s_waitcnt vmcnt(0), a wait for the next chunk's DMA rows and every store
                                // in flight (first version of this fix: the clean 16k job 267 -> 312 us)
#pragma unroll
                                for (uint32_t i = 0; i < 4; i++) asm volatile("" : "+v"(ua[i]), "+v"(ub[i]));
                            }
                        }
                        if (!is_idle && !BT_ABLATE(A, 2u)) {
#pragma unroll
                            for (uint32_t i = 0; i < 4; i++) {
                                if (BT_ABLATE(A, 8192u)) __builtin_nontemporal_store(ua[i] | (ub[i] << 16), &dst5[(4 * quad + i) * (T / 2)]);  // (8192: streaming stores)
                                else dst5[(4 * quad + i) * (T / 2)] = ua[i] | (ub[i] << 16);
                            }
                        }
                        if (kDma && kDmaPos == quad + 1 && more) dma_issue(s_buf + ((k + 1) & 1u) * buf_texels, next_ymin, next_slots);
                        if (do4) {
                            // two level-1 pixels (row pairs 0-1, 2-3 of the quad), one per packed lane: ((a0 + a1) + b0) + b1, / 4
                            const f2 sum = ((conv2(ua[0], ua[2]) + conv2(ua[1], ua[3])) + conv2(ub[0], ub[2])) + conv2(ub[1], ub[3]);
                            const f2 wq = quantise_quarter(sum);
                            q[2 * quad] = uint32_t(wq.x);
                            q[2 * quad + 1] = uint32_t(wq.y);
                            if (kFix && __builtin_expect(zq[quad] == 0, 0)) {
                                // kept texels may be 0 (no data): the valid-average (downsample.wgsl:25-39) in the scaled domain, like the tail's down_pair_r16 —
                                // a zero adds nothing to `sum` (F(0) = 0, and x + 0 is exact), so the sum of the valid texels is already there: one IEEE
                                // division by their count per pixel (0 / 1 = 0 for a block without data), no second pass over the texels
                                const uint32_t ca = min(ua[0], 1u) + min(ua[1], 1u) + min(ub[0], 1u) + min(ub[1], 1u), cb = min(ua[2], 1u) + min(ua[3], 1u) + min(ub[2], 1u) + min(ub[3], 1u);
                                const f2 d = {sum.x / float(max(ca, 1u)), sum.y / float(max(cb, 1u))};
                                const f2 wv = khalf + kn * d;
                                q[2 * quad] = uint32_t(wv.x);
                                q[2 * quad + 1] = uint32_t(wv.y);
                            }
                        }
                    }
                    if (do4) {
                        // centre + (edge columns) the x neighbour's apron column; apron rows come from the stitch launch.
                        // The lane pair (2m, 2m+1) holds two adjacent pixels: the even lane stores both as one aligned dword
                        // (b, c / 2 even) — 2-byte stores cost the memory pipeline about as much as 4-byte ones.
                        uint32_t both[4];
#pragma unroll
                        for (uint32_t j = 0; j < 4; j++)
                            both[j] = q[j] | (uint32_t(__builtin_amdgcn_update_dpp(0, int(q[j]), 0xB1, 0xf, 0xf, true)) << 16);
                        if (is_centre && (tid & 1u) == 0 && !BT_ABLATE(A, 4u)) {
                            uint32_t* dst = reinterpret_cast<uint32_t*>(tile4 + (b + cy4_first) * T + b + cx4);
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++) {
                                if (BT_ABLATE(A, 16384u)) __builtin_nontemporal_store(both[j], &dst[j * (T / 2)]);  // (16384: streaming parent stores)
                                else dst[j * (T / 2)] = both[j];
                            }
                        }
                        if (x4_count && !BT_ABLATE(A, 4u)) {  // a few lanes of a tile; one texel each unless the x neighbour is absent
                            uint16_t* t = A.atlas + uint64_t(x4_layer) * tile_texels + x4_off + cy4_first * T;
#pragma unroll
                            for (uint32_t j = 0; j < 4; j++) t[j * T] = uint16_t(q[j]);
                            if (x4_count > 1)
                                for (uint32_t e = 1; e < x4_count; e++)
#pragma unroll
                                    for (uint32_t j = 0; j < 4; j++) t[j * T + e] = uint16_t(q[j]);
                        }
                        if (do3) {
                            // the lane pair (2m, 2m+1) owns two level-2 pixels (quads 0 and 1): the even lane finishes
                            // quad 0, the odd lane quad 1.  Sum order ((x0y0 + x0y1) + x1y0) + x1y1: the even lane holds the
                            // x0 column, the odd lane x1.  Every lane converts its own four values; the even lane sends its
                            // quad-1 column sum, the odd lane its two quad-0 values (DPP swap inside the lane pair, no LDS).
                            const bool even = (tid & 1u) == 0;
                            const f2 c02 = conv2(q[0], q[2]), c13 = conv2(q[1], q[3]);
                            const f2 colsum = c02 + c13;  // {quad 0, quad 1} column sums of this lane
                            auto swap_pair = [](float v) -> float {  // quad_perm [1, 0, 3, 2]
                                return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
                            };
                            const float recv1 = swap_pair(even ? colsum.y : c02.x), recv2 = swap_pair(c13.x);
                            const float sa = even ? colsum.x : recv1, sb = even ? recv1 : c02.y, sc = even ? recv2 : c13.y;
                            const float s3 = (sa + sb) + sc;
                            uint32_t w3 = uint32_t(0.5f + (0.25f * (65535.0f / 65536.0f)) * s3);  // scaled domain, see quantise_quarter; the clamp is a no-op here
                            if constexpr (kFix) {
                                // a LOD-1 texel of the 2 x 2 block without data (0: only ever the result of a no-data quad): the valid-average
                                const uint32_t m0 = even ? q[0] : q[2], m1 = even ? q[1] : q[3];                         // this lane's column
                                const uint32_t p0 = (even ? both[0] : both[2]) >> 16, p1 = (even ? both[1] : both[3]) >> 16;  // the partner's
                                if (__builtin_expect(min(min(m0, m1), min(p0, p1)) == 0, 0)) w3 = even ? downsample4(m0, m1, p0, p1) : downsample4(p0, p1, m0, m1);
                            }
                            const uint32_t row3 = (b + cy3_first + (even ? 0u : 1u)) * T;
                            if (is_centre && !BT_ABLATE(A, 64u)) tile3[row3 + b + cx3] = uint16_t(w3);
                            if (x3_count && !BT_ABLATE(A, 64u)) {
                                uint16_t* t = A.atlas + uint64_t(x3_layer) * tile_texels + x3_off + (cy3_first + (even ? 0u : 1u)) * T;
                                t[0] = uint16_t(w3);
                                if (x3_count > 1)
                                    for (uint32_t e = 1; e < x3_count; e++) t[e] = uint16_t(w3);
                            }
                        }
                    }
                    if (BT_ABLATE(A, 1073741824u)) {
                        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the chunk barrier's own wait, a few instructions early)
#pragma unroll
                        for (uint32_t i = 0; i < 8; i++) asm volatile("" ::"v"(sink[i]));
                    }
                } else {
                uint32_t zrow = 1;  // smallest raw texel of the row hblend converted last
                auto hblend = [&](int y) -> f2 {  // (mix(t00, t10, fx) for column a, same for column b) of source row y
                    const uint16_t* row = s_src + uint32_t(y - cur_ymin) * P;
                    const uint32_t r0 = row[la0], r1 = row[lb0], r2 = row[la1], r3 = row[lb1];
                    if constexpr (kFix) zrow = min(min(r0, r1), min(r2, r3));
                    const f2 left = conv2(r0, r1), right = conv2(r2, r3);
                    return left * gx + right * fx;
                };
                f2 hcur = kzero;
                int hy = -1;
                uint32_t zcur = 1;  // ... and of the row hcur came from
                for (uint32_t q = 0; q < nrows; q += 4) {
                    uint32_t ua[4], ub[4];
                    uint32_t zq = 1;
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++) {
                        const int yy = __builtin_amdgcn_readfirstlane(row_y0[q + i]);
                        const int y0 = yy & 0x7fffffff, y1 = y0 + (yy < 0 ? 0 : 1);
                        const float fy = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, row_fy[q + i].x)));
                        f2 top = hcur;
                        uint32_t ztop = zcur;
                        if (y0 != hy) {
                            top = hblend(y0);
                            ztop = zrow;
                        }
                        f2 bot = top;
                        uint32_t zbot = ztop;
                        if (y1 != y0) {
                            bot = hblend(y1);
                            zbot = zrow;
                        }
                        hcur = bot;
                        zcur = zbot;
                        hy = y1;
                        zq = min(zq, min(ztop, zbot));
                        const f2 fy2 = {fy, fy}, gy2 = {1.0f - fy, 1.0f - fy};
                        const f2 w = quantise(top * gy2 + bot * fy2);
                        ua[i] = uint32_t(w.x);
                        ub[i] = uint32_t(w.y);
                    }
                    const bool fix = kFix && zq == 0;
                    if (fix) {
                        // (rare) this thread read a no-data texel for the quad: per-pixel validity from the staged rows, the previous
                        // atlas value where a footprint has no data (split.wgsl:34-42) — in place, like the static path above
                        const uint16_t* h = A.atlas + uint64_t(home_col) * tile_texels + (b + cr0 + q) * T + b;
#pragma unroll
                        for (uint32_t i = 0; i < 4; i++) {
                            const int yy = row_y0[q + i];
                            const uint16_t* r0 = s_src + uint32_t((yy & 0x7fffffff) - cur_ymin) * P;
                            const uint16_t* r1 = r0 + (yy < 0 ? 0u : P);
                            if (min(min(r0[la0], r0[la1]), min(r1[la0], r1[la1])) == 0) ua[i] = A.prev_zero ? 0u : uint32_t(h[i * T + rxa]);
                            if (min(min(r0[lb0], r0[lb1]), min(r1[lb0], r1[lb1])) == 0) ub[i] = A.prev_zero ? 0u : uint32_t(h[i * T + rxb]);
                        }
#pragma unroll
                        for (uint32_t i = 0; i < 4; i++) asm volatile("" : "+v"(ua[i]), "+v"(ub[i]));  // (the values arrive inside the rare branch, see the static path)
                    }
                    const uint32_t py = b + cr0 + q;
                    if (!is_idle && !BT_ABLATE(A, 2u)) {
#pragma unroll
                        for (uint32_t i = 0; i < 4; i++) tile5_u32[((py + i) * T + px0) >> 1] = ua[i] | (ub[i] << 16);
                    }
                    if (kDma && kDmaPos != 0 && q == ((kDmaPos == 1 || kDmaPos >= 10) ? 0u : (nrows > 4 ? 4u : 0u)) && more) dma_issue(s_buf + ((k + 1) & 1u) * buf_texels, next_ymin, next_slots);
                    if (do4) {
                        // two level-1 pixels (row pairs 0-1 and 2-3) in the two packed lanes: ((a0 + a1) + b0) + b1, then / 4
                        const f2 s = ((conv2(ua[0], ua[2]) + conv2(ua[1], ua[3])) + conv2(ub[0], ub[2])) + conv2(ub[1], ub[3]);
                        const f2 wq = quantise_quarter(s);
                        uint32_t q0 = uint32_t(wq.x), q1 = uint32_t(wq.y);
                        if (fix) {  // kept texels may be 0 (no data): the valid-average (downsample.wgsl:25-39) — the scaled sum divided by the count, see the static path
                            const uint32_t ca = min(ua[0], 1u) + min(ua[1], 1u) + min(ub[0], 1u) + min(ub[1], 1u), cb = min(ua[2], 1u) + min(ua[3], 1u) + min(ub[2], 1u) + min(ub[3], 1u);
                            const f2 d = {s.x / float(max(ca, 1u)), s.y / float(max(cb, 1u))};
                            const f2 wv = khalf + kn * d;
                            q0 = uint32_t(wv.x);
                            q1 = uint32_t(wv.y);
                        }
                        const uint32_t cy = cr0 + q, cy4 = cy4_base + (cy >> 1);
                        if (is_centre) {
                            uint16_t* dst = tile4 + (b + cy4) * T + b + cx4;
                            dst[0] = uint16_t(q0);
                            dst[T] = uint16_t(q1);
                        }
                        if (x4_count) {
                            xpush4(cy4, uint16_t(q0));
                            xpush4(cy4 + 1, uint16_t(q1));
                        }
                        if (do3) {
                            const uint32_t other0 = __shfl_xor(q0, 1), other1 = __shfl_xor(q1, 1);
                            if (is_centre && (tid & 1u) == 0) {
                                const f2 mine = conv2(q0, q1), theirs = conv2(other0, other1);
                                const float s3 = ((mine.x + mine.y) + theirs.x) + theirs.y;
                                uint32_t w3 = uint32_t(0.5f + (0.25f * (65535.0f / 65536.0f)) * s3);  // scaled domain (conv2)
                                // a LOD-1 texel of the block without data (0: only ever the result of a no-data quad): the valid-average
                                if (kFix && min(min(q0, q1), min(other0, other1)) == 0) w3 = downsample4(q0, q1, other0, other1);
                                const uint32_t cy3 = cy3_base + (cy >> 2);
                                tile3[(b + cy3) * T + b + cx3] = uint16_t(w3);
                                if (x3_count) xpush3(cy3, uint16_t(w3));
                            }
                        }
                    }
                }
                }
            } else {
                generic_rows(k);
            }
        }

        if (!more) break;
        // chunk k + 1 goes into the other staging buffer (nobody reads it any more: its last readers passed the
        // previous barrier), the row table of chunk k + 3 replaces the one of chunk k after the barrier
        if (kStaged && !kDma && !BT_ABLATE(A, 8u)) {
            uint16_t* s_next = s_buf + ((k + 1) & 1u) * buf_texels;
            if (wide) stage_commit(s_next, next_slots, pre);
            else stage_narrow(s_next, next_ymin, next_slots);
        }
        ymin = next_ymin;
        slots = next_slots;
        if constexpr (kDma && kP == 0) {
            if (one_buffer) {  // every wave is through with this chunk's rows: the next chunk's travel into the same buffer
                __syncthreads();
                dma_issue(s_buf, next_ymin, next_slots);
            }
        }
        chunk_barrier();
    }
    wg_stamp(2);
#ifdef BT_DEBUG_HOOKS
    if (BT_ABLATE(A, 134217728u) && tid == 0)
        reinterpret_cast<unsigned long long*>(A.atlas + uint64_t(A.m.atlas_size - 1u) * tile_texels)[item_index * 8u + 4u] = __builtin_amdgcn_s_memrealtime();
#endif
    wg_stamp(3);
}

// fast / non-staged variants: workgroup = (tile, part of its chunks), XCD-contiguous order
template <bool kStaged, bool kGeneric, uint32_t kT, uint32_t kP, bool kDma = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void fused_main_kernel(FusedArgs A) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const uint32_t work = BT_ABLATE(A, 1024u) ? blockIdx.x : xcd_remap(blockIdx.x, gridDim.x);  // (1024: dispatch order, no XCD remap)
    const uint32_t chunks_per_tile = (A.m.center_size + kMainRows - 1) / kMainRows;
    const uint32_t part = work % A.groups;
    const uint32_t k_begin = part * chunks_per_tile / A.groups, k_end = (part + 1) * chunks_per_tile / A.groups;
    if (k_begin >= k_end) return;  // more parts than chunks (tiny tiles)
#ifdef BT_DEBUG_HOOKS
#define BT_FUSED_DEBUG_ENTRY_PRIORITY
#include "bt_fused_debug.inc"  // (static wave priority by dispatch rank: timing experiment)
#undef BT_FUSED_DEBUG_ENTRY_PRIORITY
#endif
    fused_main_chunks<kStaged, kGeneric, kT, kP, kDma>(A, work / A.groups, k_begin, k_end, smem);
}

__global__ __launch_bounds__(64) void fused_corner_kernel(FusedArgs A) { corner_pixels(A, blockIdx.x, threadIdx.x, blockDim.x); }

// ---- fused_tail: up to three LODs below `A.lod`, read from the atlas ---------------------------------
// Workgroup = 64 x 64 pixels of the LOD-`lod` mosaic, 16 x 16 threads, 4 x 4 input pixels per thread: a thread
// owns 2 x 2 pixels of lod-1 and one pixel of lod-2 in registers; 2 x 2 neighbouring threads (lanes l, l+1,
// l+16, l+17 of one wave) combine into one pixel of lod-3 with three shuffles.  No LDS, no barrier, one tile
// lookup per thread and LOD (c % 4 == 0: a 4 x 4 block never straddles a tile).
// Top / bottom apron rows (whole rows, corners included) of the tiles of the LODs fused_main produced below the
// finest one: stitch.wgsl:53-118 for same-side neighbours — neighbour's centre rows, or the own centre clamped
// when it is absent.  (Cube face edges are re-stitched afterwards by the generic kernel.)  One thread per pixel pair.
constexpr uint32_t kApronPairsPerThread = 4;  // (a workgroup = 1024 pixel pairs: a T = 512, b = 2 tile's four apron rows — a quarter of the extra workgroups of one pair per thread)
__device__ __forceinline__ void tail_apron_rows(const FusedArgs& A, uint32_t side, uint32_t e) {
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size, o = b + c;
    const uint32_t pairs = b * T, per_block = 256u * kApronPairsPerThread, blocks_per_tile = (pairs + per_block - 1u) / per_block;
    for (uint32_t k = 0; k < A.apron_lods; k++) {
        const uint32_t lod = A.lod + k, n = 1u << lod, blocks = n * n * blocks_per_tile;
        if (e >= blocks) {
            e -= blocks;
            continue;
        }
        const uint32_t tile = e / blocks_per_tile, i0 = (e % blocks_per_tile) * per_block + threadIdx.x;
        const uint32_t tx = tile / n, ty = tile % n;
        const uint32_t self = grid_lookup(A, side, lod, int(tx), int(ty));
        if (self == kInvalid) return;
        // every load first (unconditional, from clamped addresses), then the stores: one round trip for the thread's four pairs
        uint32_t v[kApronPairsPerThread][2], dst[kApronPairsPerThread];
        bool live[kApronPairsPerThread];
#pragma unroll
        for (uint32_t q = 0; q < kApronPairsPerThread; q++) {
            const uint32_t i = i0 + 256u * q;
            live[q] = i < pairs;
            const uint32_t ii = live[q] ? i : 0u;
            const uint32_t r = ii / (T / 2u), px = 2u * (ii % (T / 2u)), py = r < b ? r : c + r;
            const int rx = px < b ? -1 : (px >= o ? 1 : 0), ry = r < b ? -1 : 1;
            if (A.seam_skip) {
                // cube: a neighbour beyond ONE edge of the face lives on another face and a seam workgroup of this launch writes the region
                // (beyond two edges — the cube's corner — there is none: clamped below like any absent neighbour)
                const bool out_x = int(tx) + rx < 0 || int(tx) + rx >= int(n), out_y = int(ty) + ry < 0 || int(ty) + ry >= int(n);
                if (out_x != out_y) live[q] = false;
            }
            const uint32_t nb = grid_lookup(A, side, lod, int(tx) + rx, int(ty) + ry);
#pragma unroll
            for (uint32_t h = 0; h < 2; h++) {
                const uint32_t x = px + h;
                const uint32_t sx = nb != kInvalid ? uint32_t(int(x) - rx * int(c)) : min(max(x, b), o - 1u);
                const uint32_t sy = nb != kInvalid ? uint32_t(int(py) - ry * int(c)) : min(max(py, b), o - 1u);
                v[q][h] = A.atlas[uint64_t(nb != kInvalid ? nb : self) * T * T + sy * T + sx];
            }
            dst[q] = py * T + px;
        }
#pragma unroll
        for (uint32_t q = 0; q < kApronPairsPerThread; q++)
            if (live[q]) *reinterpret_cast<uint32_t*>(A.atlas + uint64_t(self) * T * T + dst[q]) = v[q][0] | (v[q][1] << 16);
        return;
    }
}

// Rgba8 twin (one 4-byte texel per thread) that also does the left / right apron columns: after fused_direct, which writes
// the centres of the two parent LODs only, these workgroups are the whole stitch of those tiles (stitch.wgsl:53-118 for
// same-face neighbours; cube seams are re-stitched by the batched kernel afterwards, like everywhere in the fused plans).
__device__ __forceinline__ uint32_t tail_apron_texels_per_tile(const FusedArgs& A) {
    return 2u * A.m.border_size * A.m.texture_size + (A.apron_cols ? 2u * A.m.border_size * A.m.center_size : 0u);
}
__device__ __forceinline__ void tail_aprons_rgba8(const FusedArgs& A, uint32_t side, uint32_t e) {
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size, o = b + c;
    const uint32_t per_tile = tail_apron_texels_per_tile(A), blocks_per_tile = (per_tile + 255u) / 256u;
    uint32_t* atlas = reinterpret_cast<uint32_t*>(A.atlas);
    for (uint32_t k = 0; k < A.apron_lods; k++) {
        const uint32_t lod = A.lod + k, n = 1u << lod, blocks = n * n * blocks_per_tile;
        if (e >= blocks) {
            e -= blocks;
            continue;
        }
        const uint32_t tile = e / blocks_per_tile, i = (e % blocks_per_tile) * 256u + threadIdx.x;
        if (i >= per_tile) return;
        const uint32_t tx = tile / n, ty = tile % n;
        const uint32_t self = grid_lookup(A, side, lod, int(tx), int(ty));
        if (self == kInvalid) return;
        uint32_t px, py;
        if (i < 2u * b * T) {  // whole apron rows (with the corners)
            const uint32_t r = i / T;
            px = i % T;
            py = r < b ? r : c + r;
        } else {  // apron columns of the centre rows
            const uint32_t j = i - 2u * b * T, kk = j % (2u * b);
            px = kk < b ? kk : c + kk;
            py = b + j / (2u * b);
        }
        const int rx = px < b ? -1 : (px >= o ? 1 : 0), ry = py < b ? -1 : (py >= o ? 1 : 0);
        if (A.seam_skip) {  // cube: a region beyond exactly ONE face edge belongs to a seam workgroup of this launch (see tail_apron_rows)
            const bool out_x = int(tx) + rx < 0 || int(tx) + rx >= int(n), out_y = int(ty) + ry < 0 || int(ty) + ry >= int(n);
            if (out_x != out_y) return;
        }
        const uint32_t nb = grid_lookup(A, side, lod, int(tx) + rx, int(ty) + ry);
        const uint32_t sx = nb != kInvalid ? uint32_t(int(px) - rx * int(c)) : min(max(px, b), o - 1u);
        const uint32_t sy = nb != kInvalid ? uint32_t(int(py) - ry * int(c)) : min(max(py, b), o - 1u);
        atlas[uint64_t(self) * T * T + py * T + px] = atlas[uint64_t(nb != kInvalid ? nb : self) * T * T + sy * T + sx];
        return;
    }
}

// kRegular: the atlas indices follow the closed form (FusedArgs::regular) — as a compile-time fact the table lookups and their
// registers fall away (64 VGPRs: eight waves per SIMD)
// downsample.wgsl:25-39 for R16 in the 2^16-scaled domain of fused_main's fast loop (round 5; downsample4 above is the plain form).
// F(t) = fma(x, r, x) = 65536 * RN(t / 65535) and F(0) = 0: a no-data texel adds an exact 0 to the running sum, so ((F00 + F01) + F10) + F11
// IS the sum over the valid texels in OFFSETS order, rounding for rounding; what differs is the divisor.  All four valid (the common case):
// 0.5 + (0.25 * k) * sum with k = 65535 / 65536 (sum * 0.25 is exact).  Otherwise ONE IEEE division by the count — scaling by a power of two
// commutes with it, and an average of values <= 1 needs no clamp — and count 0 gives 0.5 + k * (0 / 1) -> 0, the defined "no data".
typedef float tail_f2 __attribute__((ext_vector_type(2)));
typedef uint16_t tail_u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ tail_f2 tail_conv2(uint32_t a, uint32_t bq) {
    const tail_f2 x = {float(a), float(bq)}, kr = {1.0f / 65535.0f, 1.0f / 65535.0f};
    return __builtin_elementwise_fma(x, kr, x);
}
// two 2 x 2 averages at once: block A = a_top over a_bot, packed texel pairs (low half = x0), block B likewise
__device__ __forceinline__ void down_pair_r16(uint32_t a_top, uint32_t a_bot, uint32_t b_top, uint32_t b_bot, uint32_t& qa, uint32_t& qb) {
    const tail_f2 khalf = {0.5f, 0.5f}, kn = {65535.0f / 65536.0f, 65535.0f / 65536.0f}, knq = {0.25f * (65535.0f / 65536.0f), 0.25f * (65535.0f / 65536.0f)};
    const tail_u16x2 one = {1, 1};
    const tail_u16x2 at = __builtin_bit_cast(tail_u16x2, a_top), ab = __builtin_bit_cast(tail_u16x2, a_bot), bt2 = __builtin_bit_cast(tail_u16x2, b_top), bb = __builtin_bit_cast(tail_u16x2, b_bot);
    const tail_u16x2 m = __builtin_elementwise_min(__builtin_elementwise_min(at, ab), __builtin_elementwise_min(bt2, bb));
    // OFFSETS order (0,0),(0,1),(1,0),(1,1) of (dx, dy): ((x0y0 + x0y1) + x1y0) + x1y1
    const tail_f2 sum = ((tail_conv2(a_top & 0xFFFFu, b_top & 0xFFFFu) + tail_conv2(a_bot & 0xFFFFu, b_bot & 0xFFFFu)) + tail_conv2(a_top >> 16, b_top >> 16)) + tail_conv2(a_bot >> 16, b_bot >> 16);
    tail_f2 w = khalf + knq * sum;
    if (__builtin_expect(m.x == 0 || m.y == 0, 0)) {  // (rare) some texel has no data: the valid-average
        const tail_u16x2 ca = __builtin_elementwise_min(at, one) + __builtin_elementwise_min(ab, one), cb = __builtin_elementwise_min(bt2, one) + __builtin_elementwise_min(bb, one);
        const tail_f2 d = {sum.x / float(max(uint32_t(ca.x) + uint32_t(ca.y), 1u)), sum.y / float(max(uint32_t(cb.x) + uint32_t(cb.y), 1u))};
        w = khalf + kn * d;
    }
    qa = uint32_t(w.x);
    qb = uint32_t(w.y);
}
__device__ __forceinline__ uint32_t down_one_r16(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11) {
    const float r = 1.0f / 65535.0f;
    const float x00 = float(t00), x01 = float(t01), x10 = float(t10), x11 = float(t11);
    const float sum = ((__builtin_fmaf(x00, r, x00) + __builtin_fmaf(x01, r, x01)) + __builtin_fmaf(x10, r, x10)) + __builtin_fmaf(x11, r, x11);
    float w = 0.5f + (0.25f * (65535.0f / 65536.0f)) * sum;
    if (__builtin_expect(min(min(t00, t01), min(t10, t11)) == 0, 0)) {
        const uint32_t count = min(t00, 1u) + min(t01, 1u) + min(t10, 1u) + min(t11, 1u);
        w = 0.5f + (65535.0f / 65536.0f) * (sum / float(max(count, 1u)));
    }
    return uint32_t(w);
}

// Pixel (mx, my) of the LOD-`lod` mosaic of face `side`, evaluated from the tail's INPUT LOD (lod + K, complete when the launch starts) with the
// tail's own reduction in its own order — (x, y), (x, y + 1), (x + 1, y), (x + 1, y + 1), downsample.wgsl:25-39 per level, every level quantised —
// i.e. bit for bit what the mosaic workgroups of this launch write into that tile's centre (R16).
template <typename TT>
__device__ __forceinline__ uint32_t tail_down(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11) {  // the tail's reduction of one level, per format
    if constexpr (std::is_same<TT, uint16_t>::value) return down_one_r16(t00, t01, t10, t11);
    else return downsample4_rgba8(t00, t01, t10, t11);
}
template <int K, typename TT>
__device__ __forceinline__ uint32_t pull_value(const FusedArgs& A, uint32_t side, uint32_t lod, uint32_t mx, uint32_t my) {
    if constexpr (K == 0) {
        const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size;
        const uint32_t tx = mx / c, ty = my / c;
        const uint32_t idx = grid_lookup(A, side, lod, int(tx), int(ty));
        if (idx == kInvalid) return 0u;  // an absent tile reads as no data, like the mosaic workgroups' loads
        return reinterpret_cast<const TT*>(A.atlas)[uint64_t(idx) * T * T + uint64_t(b + my - ty * c) * T + b + mx - tx * c];
    } else {
        const uint32_t t00 = pull_value<K - 1, TT>(A, side, lod + 1u, 2u * mx, 2u * my), t01 = pull_value<K - 1, TT>(A, side, lod + 1u, 2u * mx, 2u * my + 1u);
        const uint32_t t10 = pull_value<K - 1, TT>(A, side, lod + 1u, 2u * mx + 1u, 2u * my), t11 = pull_value<K - 1, TT>(A, side, lod + 1u, 2u * mx + 1u, 2u * my + 1u);
        return tail_down<TT>(t00, t01, t10, t11);
    }
}

// ONE cross-face apron region of a tile the tail itself produces (stitch.wgsl:12-51, 79-118), pulled: the texel the reference copies out of
// the neighbour face's centre is evaluated from the tail's input instead (task.raster = LODs between the tile and the input; the neighbour
// tile's coordinate rides in rel_index[region] as x << 16 | y).  One workgroup of 256 threads; kPack = 2: two adjacent pixels per dword store.
template <typename TT, uint32_t kPack>
__device__ __forceinline__ void stitch_region_pull(const FusedArgs& A, const TaskDev& task) {
    const uint32_t Tsz = A.m.texture_size, b = A.m.border_size, c = A.m.center_size, o = b + c;
    const uint32_t region = uint32_t(__ffs(int(task.regions))) - 1u;  // 0 top, 1 right, 2 bottom, 3 left, 4 TL, 5 TR, 6 BR, 7 BL
    const uint32_t x0 = (region == 0u || region == 2u) ? b : ((region == 1u || region == 5u || region == 6u) ? o : 0u);
    const uint32_t y0 = (region == 1u || region == 3u) ? b : ((region == 2u || region == 6u || region == 7u) ? o : 0u);
    const uint32_t w = (region == 0u || region == 2u) ? c : b, h = (region == 1u || region == 3u) ? c : b;
    const int rx = (region == 1u || region == 5u || region == 6u) ? 1 : ((region == 3u || region == 4u || region == 7u) ? -1 : 0);
    const int ry = (region == 2u || region == 6u || region == 7u) ? 1 : ((region == 0u || region == 4u || region == 5u) ? -1 : 0);
    const uint32_t other = task.rel_side[region], nx = task.rel_index[region] >> 16, ny = task.rel_index[region] & 0xFFFFu;
    // task.raster = levels | part << 8 | parts << 16: a region is shared out over `parts` workgroups (a thread then evaluates one pixel pair:
    // the pull workgroups are the longest of the launch — 4^levels dependent-free loads per pixel — and must not outlast the mosaic's)
    const uint32_t levels = task.raster & 0xFFu, part = (task.raster >> 8) & 0xFFu, parts = max(1u, task.raster >> 16);
    const uint32_t count = (w / kPack) * h, share = (count + parts - 1u) / parts, i_end = min(count, (part + 1u) * share);
    // levels <= 2: the 2^levels x 2^levels input block of a pixel lies in ONE input tile (c is a multiple of 4): one tile lookup, one division per axis
    const uint32_t in_lod = task.lod + levels, cn = c >> levels;  // cn: pixels of this LOD one input tile's centre covers
    for (uint32_t i = part * share + threadIdx.x; i < i_end; i += 256u) {
        const uint32_t px = x0 + kPack * (i % (w / kPack)), py = y0 + i / (w / kPack);
        if (py >= A.m.row_limit) continue;
        uint32_t v[kPack];
#pragma unroll
        for (uint32_t e = 0; e < kPack; e++) {
            // neighbour_data of stitch.wgsl: the apron texel as a texel of the neighbour tile (its centre: [b, b + c) on both axes), then as a pixel of its face's mosaic
            const uint2 q = project_to_side(uint32_t(int(px + e) - rx * int(c)), uint32_t(int(py) - ry * int(c)), Tsz, task.side, other);
            const uint32_t mx = nx * c + (q.x - b), my = ny * c + (q.y - b);
            if (BT_ABLATE(A, 2147483648u)) {  // (2147483648: pull workgroups store zeros without evaluating anything — timing experiment)
                v[e] = 0;
            } else if (levels <= 2u && (c & 3u) == 0) {
                const uint32_t tx = mx / cn, ty = my / cn;  // the input tile, and the block's first pixel in it
                const uint32_t idx = grid_lookup(A, other, in_lod, int(tx), int(ty));
                const TT* src = reinterpret_cast<const TT*>(A.atlas) + uint64_t(idx == kInvalid ? 0u : idx) * Tsz * Tsz + uint64_t(b + ((my - ty * cn) << levels)) * Tsz + b + ((mx - tx * cn) << levels);
                // every load unconditional and issued before the first use (a conditional load becomes its own divergent block with a wait behind it:
                // sixteen dependent round trips per pixel made these the longest workgroups of the launch); an absent tile reads as no data
                const uint32_t keep = idx == kInvalid ? 0u : 0xFFFFFFFFu;
                if (levels == 1u) {
                    const uint32_t t00 = src[0], t01 = src[Tsz], t10 = src[1], t11 = src[Tsz + 1u];
                    v[e] = tail_down<TT>(t00 & keep, t01 & keep, t10 & keep, t11 & keep);
                } else {
                    uint32_t t[4][4];  // [dy][dx]
#pragma unroll
                    for (uint32_t dy = 0; dy < 4; dy++)
#pragma unroll
                        for (uint32_t dx = 0; dx < 4; dx++) t[dy][dx] = uint32_t(src[dy * Tsz + dx]) & keep;
                    auto two = [&](uint32_t dx, uint32_t dy) -> uint32_t { return tail_down<TT>(t[dy][dx], t[dy + 1][dx], t[dy][dx + 1], t[dy + 1][dx + 1]); };
                    v[e] = tail_down<TT>(two(0u, 0u), two(0u, 2u), two(2u, 0u), two(2u, 2u));
                }
            } else {
                v[e] = levels == 1u ? pull_value<1, TT>(A, other, task.lod, mx, my) : levels == 2u ? pull_value<2, TT>(A, other, task.lod, mx, my) : pull_value<3, TT>(A, other, task.lod, mx, my);
            }
        }
        TT* dst = reinterpret_cast<TT*>(A.atlas) + uint64_t(task.atlas_index) * Tsz * Tsz + uint64_t(py) * Tsz + px;
        if (BT_ABLATE(A, 4194304u) && (v[0] | v[kPack - 1]) != 0x12345u) continue;  // (4194304: evaluated, not stored — timing experiment)
        if constexpr (kPack == 2) *reinterpret_cast<uint32_t*>(dst) = v[0] | (v[1] << 16);
        else *dst = TT(v[0]);
    }
}

template <uint32_t kFormat, bool kRegular>
__global__ __launch_bounds__(256) void fused_tail_kernel(FusedArgs A_in) {
    FusedArgs A = A_in;
    A.regular = kRegular ? 1u : 0u;
    constexpr bool kR16 = kFormat == BT_FORMAT_R16;
    using TT = typename std::conditional<kR16, uint16_t, uint32_t>::type;  // texel
    // Workgroup -> work, XCD-aware (round 5).  Workgroups go to the XCDs round robin (blockIdx.x % 8) and every XCD has its own L2; a
    // 64-pixel-wide mosaic column starts b texels into a tile row, i.e. it shares its first and last 128-byte line with the workgroup
    // beside it — dispatched in mosaic order the two sit on different XCDs and every line is fetched from HBM twice
    // (profiles/r05_pmc_summary.json: the tail read 67.7 MB for 33.5 MB of texels).  So: XCD k takes the k-th eighth of the mosaic
    // workgroups (whole rows of them, neighbours in x back to back on one L2), and in front of those its eighth of the extra workgroups —
    // apron rows (Rgba8: and columns) of the LODs above, on a cube the cross-face seam regions first: short dependent chains that run
    // beside the mosaic work instead of behind the last of it.
    // The extras of an XCD, in dispatch order: its share of the cross-face seam regions (task f = j * 8 + xcd: the list's head — the pulled
    // regions, the launch's longest workgroups — spreads over all eight XCDs; round 6: with all of them on XCD 0 the launch ended 4 us late),
    // then its eighth of the apron blocks (side-major), then its eighth of the mosaic.
    const uint32_t nx = ((1u << A.lod) * A.m.center_size + 63u) / 64u, per_side_m = nx * nx, per_side_a = A.tail_extras;
    const uint32_t total_m = A.sides * per_side_m, total_a = A.sides * per_side_a, chunk_m = (total_m + 7u) / 8u, chunk_a = (total_a + 7u) / 8u;
    const uint32_t chunk_s = (A.seam_count + 7u) / 8u, chunk_e = chunk_s + chunk_a;
    uint32_t side, block_x, block_y;
    {
        const uint32_t xcd = blockIdx.x % 8u, j = blockIdx.x / 8u;
        if (j < chunk_e) {
            if (BT_ABLATE(A, 268435456u)) return;  // (268435456: no extra workgroups — timing experiment)
            if (j < chunk_s) {  // one cross-face region
                const uint32_t f = j * 8u + xcd;
                if (f < A.seam_count) {
                    const bool pulled = A.seam_tasks[f].raster != 0u;  // a region of a tile this launch produces: pulled from the input LOD
                    if constexpr (kR16) {
                        const bool pairs = (A.m.border_size & 1u) == 0 && (A.m.texture_size & 1u) == 0;
                        if (pulled) {
                            if (pairs) stitch_region_pull<uint16_t, 2>(A, A.seam_tasks[f]);
                            else stitch_region_pull<uint16_t, 1>(A, A.seam_tasks[f]);
                        } else if (pairs) stitch_region_body<uint16_t, 2>(A.m, A.atlas, A.seam_tasks[f]);
                        else stitch_region_body<uint16_t, 1>(A.m, A.atlas, A.seam_tasks[f]);
                    } else {
                        if (pulled) stitch_region_pull<uint32_t, 1>(A, A.seam_tasks[f]);
                        else stitch_region_body<uint32_t, 1>(A.m, A.atlas, A.seam_tasks[f]);
                    }
                }
                return;
            }
            const uint32_t id = xcd * chunk_a + (j - chunk_s);
            if (id >= total_a) return;
            side = id / per_side_a;
            const uint32_t e = id - side * per_side_a;
            if constexpr (kR16) tail_apron_rows(A, side, e);
            else tail_aprons_rgba8(A, side, e);
            return;
        }
        const uint32_t id = xcd * chunk_m + (j - chunk_e);
        if (id >= total_m) return;
        side = id / per_side_m;
        const uint32_t r = id - side * per_side_m;
        block_y = r / nx;
        block_x = r - block_y * nx;
    }
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size;
    const uint32_t tile_texels = T * T;
    const uint32_t size = (1u << A.lod) * c;  // mosaic extent of the input LOD (a multiple of 4)
    const uint32_t tx = threadIdx.x & 15u, ty = threadIdx.x >> 4;
    const uint32_t gx = block_x * 64u + 4u * tx, gy = block_y * 64u + 4u * ty;  // first input pixel
    const bool active = gx < size && gy < size;
    // ONE division per axis: c is a multiple of 4, so the 4 x 4 block lies in one tile, and the tile / in-tile coordinates
    // of its pixel at LOD-k follow by shifts: tile >> k, ((tile & (2^k - 1)) * c + rem) >> k
    const uint32_t tile_x = gx / c, tile_y = gy / c, rem_x = gx - tile_x * c, rem_y = gy - tile_y * c;
    const uint32_t rx1 = ((tile_x & 1u) * c + rem_x) >> 1, ry1 = ((tile_y & 1u) * c + rem_y) >> 1;
    TT* atlas = reinterpret_cast<TT*>(A.atlas);
    auto down = [](uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11) -> uint32_t {
        if constexpr (kR16) return down_one_r16(t00, t01, t10, t11);
        else return downsample4_rgba8(t00, t01, t10, t11);
    };

    uint32_t t[4][4];  // [row][col]  (R16: t[r][0], t[r][1] hold the row's two DWORDS — pixels 0 | 1 and 2 | 3 — as loaded)
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int k = 0; k < 4; k++) t[r][k] = 0;
    // all tile lookups up front (independent of the data): one round of memory latency instead of four
    uint32_t idx = kInvalid, self1 = kInvalid, self2 = kInvalid, self3 = kInvalid;
    if (active) {
        idx = grid_lookup(A, side, A.lod, int(tile_x), int(tile_y));
        self1 = grid_lookup(A, side, A.lod - 1, int(tile_x >> 1), int(tile_y >> 1));
        if (A.levels >= 2) self2 = grid_lookup(A, side, A.lod - 2, int(tile_x >> 2), int(tile_y >> 2));
        if (A.levels >= 3) self3 = grid_lookup(A, side, A.lod - 3, int(tile_x >> 3), int(tile_y >> 3));
    }
    // ... and the neighbour tiles the four lod-1 pixels will be pushed into (edge pixels only), while nothing has been stored yet.
    // b even: the 2 x 2 block (even coordinates) lies in one edge region and shares its targets; b odd: per pixel, push_pixel
    const bool pre = (b & 1u) == 0;
    const PushNb nb1 = push_targets(A, side, A.lod - 1, tile_x >> 1, tile_y >> 1, rx1, ry1, active && pre);
    if (active) {
        if (idx != kInvalid && !BT_ABLATE(A, 33554432u)) {  // an absent tile reads as no data  (33554432: no texel loads — timing experiment)
            const TT* p = atlas + uint64_t(idx) * tile_texels + (b + rem_y) * T + b + rem_x;
#pragma unroll
            for (int r = 0; r < 4; r++) {
                if constexpr (kR16) {  // 4-byte aligned (b even)
                    const uint2 w = *reinterpret_cast<const uint2 __attribute__((aligned(4)))*>(p + r * T);  // one 8-byte load from a dword-aligned address
                    t[r][0] = w.x;
                    t[r][1] = w.y;
                } else {  // 8-byte aligned (b even or not: (b + 4k) texels of 4 bytes; pairs need b even) — two texels per load
                    if ((b & 1u) == 0) {
                        const uint2 lo = *reinterpret_cast<const uint2*>(p + r * T), hi = *reinterpret_cast<const uint2*>(p + r * T + 2);
                        t[r][0] = lo.x;
                        t[r][1] = lo.y;
                        t[r][2] = hi.x;
                        t[r][3] = hi.y;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; k++) t[r][k] = p[r * T + k];
                    }
                }
            }
        }
    }
    // lod-1: 2 x 2 pixels, each from a 2 x 2 block in OFFSETS order (0,0),(0,1),(1,0),(1,1) of (dx, dy)
    uint32_t q[2][2];  // [row][col]
#pragma unroll
    for (int r = 0; r < 2; r++) {
        if constexpr (kR16) {  // the row pair's two pixels at once, from the packed dwords
            down_pair_r16(t[2 * r][0], t[2 * r + 1][0], t[2 * r][1], t[2 * r + 1][1], q[r][0], q[r][1]);
        } else {
#pragma unroll
            for (int k = 0; k < 2; k++) q[r][k] = down(t[2 * r][2 * k], t[2 * r + 1][2 * k], t[2 * r][2 * k + 1], t[2 * r + 1][2 * k + 1]);
        }
    }
    if (active) {
        const uint32_t self = self1;
        if (self != kInvalid && !BT_ABLATE(A, 1073741824u)) {  // (1073741824: no lod-1 stores — timing experiment)
            // R16: the two pixels of a row are one aligned dword of the tile (x1, b, c even); aprons per pixel, edge pixels only
            TT* centre = atlas + uint64_t(self) * tile_texels + (b + ry1) * T + b + rx1;
#pragma unroll
            for (int r = 0; r < 2; r++) {
                if constexpr (kR16) {
                    *reinterpret_cast<uint32_t*>(centre + r * T) = q[r][0] | (q[r][1] << 16);
                } else {
                    centre[r * T] = q[r][0];
                    centre[r * T + 1] = q[r][1];
                }
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    if (BT_ABLATE(A, 536870912u)) continue;  // (536870912: no apron pushes of lod-1 — timing experiment)
                    if (pre) push_store<TT>(A, nb1, self, rx1 + k, ry1 + r, TT(q[r][k]));
                    else push_pixel<false, TT>(A, side, A.lod - 1, tile_x >> 1, tile_y >> 1, self, rx1 + k, ry1 + r, TT(q[r][k]));
                }
            }
        }
    }
    if (A.levels < 2 || BT_ABLATE(A, 67108864u)) return;  // (67108864: lod-1 only — timing experiment)
    const uint32_t v2 = down(q[0][0], q[1][0], q[0][1], q[1][1]);
    if (active) {
        const uint32_t self = self2;
        if (self != kInvalid)
            push_pixel<true, TT>(A, side, A.lod - 2, tile_x >> 2, tile_y >> 2, self, ((tile_x & 3u) * c + rem_x) >> 2, ((tile_y & 3u) * c + rem_y) >> 2, TT(v2));
    }
    if (A.levels < 3) return;
    // lod-3: threads (tx, ty) with both even own the pixel; partners are lanes +1 (dx), +16 (dy), +17
    const uint32_t right = __shfl_down(v2, 1), below = __shfl_down(v2, 16), diag = __shfl_down(v2, 17);
    if (active && ((tx | ty) & 1u) == 0) {
        const uint32_t v3 = down(v2, below, right, diag);
        const uint32_t self = self3;
        if (self != kInvalid)
            push_pixel<true, TT>(A, side, A.lod - 3, tile_x >> 3, tile_y >> 3, self, ((tile_x & 7u) * c + rem_x) >> 3, ((tile_y & 7u) * c + rem_y) >> 3, TT(v3));
    }
}



// ---- fused_direct (Rgba8): split + the two parent LODs WITHOUT LDS staging ------------------------------------------
// Workgroup = several 4-row blocks of one finest tile (c = 508 = 127 x 4: no partial block), thread = one centre column (two
// sweeps of 256).  The 5 source rows x 2 texels a column needs per block are requested ONE BLOCK AHEAD straight from global
// memory (a 4-byte texel needs no sub-dword extraction, neighbouring lanes share their texels through L1), filtered
// horizontally once each; row pairs reduce in registers and lane pairs / quads through DPP to LOD-1 and LOD-2, whose centres are
// written into the parent tiles (their aprons come from the tail launch).  Validity is handled in line: a pixel without data is
// not stored (it keeps the atlas value, split.wgsl:37-42) and its previous value is fetched for the reduction.  The tile's own
// apron pixels (4 columns per row, whole apron rows in the first / last block) take the general per-pixel path with the
// neighbour tile's formula.
__device__ __forceinline__ uint32_t float_to_unorm8(float e) {
    const float cl = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
    return uint32_t(floorf(0.5f + 255.0f * cl));
}
struct H4 {
    float h[4];
    bool valid;
};
__device__ __forceinline__ H4 hrow_rgba8(uint32_t t0, uint32_t t1, float fx) {
    H4 o;
    o.valid = (t0 & 0xFFu) != 0 && (t1 & 0xFFu) != 0;  // textureGather(0, ..): channel 0 of both texels
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) o.h[k] = mixf(unorm8_to_float((t0 >> (8 * k)) & 0xFFu), unorm8_to_float((t1 >> (8 * k)) & 0xFFu), fx);
    return o;
}
__device__ __forceinline__ uint32_t vmix_rgba8(const H4& top, const H4& bot, float fy) {
    uint32_t out = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) out |= float_to_unorm8(mixf(top.h[k], bot.h[k], fy)) << (8 * k);
    return out;
}
// The same two steps in the 2^8-scaled domain (the 8-bit twin of fused_main's trick, see there): F(t) = fma(x, r, x) with
// r = RN(1 / 255) equals 256 * RN(t / 255) for all 256 inputs (bt_selftest, tests), every later operation is homogeneous,
// 255 * v == (255 / 256) * V exactly, and with inputs in [0, 1] and weights in [0, 1] the clamp of pack4x8unorm cannot
// change the result — bit-identical texels for a third of the conversion work.
__device__ __forceinline__ H4 hrow_rgba8_scaled(uint32_t t0, uint32_t t1, float fx) {
    H4 o;
    o.valid = (t0 & 0xFFu) != 0 && (t1 & 0xFFu) != 0;
    const float gx = 1.0f - fx;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) {
        const float a = float((t0 >> (8 * k)) & 0xFFu), bb = float((t1 >> (8 * k)) & 0xFFu);
        o.h[k] = __builtin_fmaf(a, 1.0f / 255.0f, a) * gx + __builtin_fmaf(bb, 1.0f / 255.0f, bb) * fx;
    }
    return o;
}
__device__ __forceinline__ uint32_t vmix_rgba8_scaled(const H4& top, const H4& bot, float fy) {
    const float gy = 1.0f - fy;
    uint32_t out = 0;
#pragma unroll
    for (uint32_t k = 0; k < 4; k++) out |= uint32_t(0.5f + (255.0f / 256.0f) * (top.h[k] * gy + bot.h[k] * fy)) << (8 * k);
    return out;
}
typedef const uint8_t __attribute__((address_space(1))) * global_bytes_t;
typedef const uint32_t __attribute__((address_space(1))) * global_u32_t;

// general evaluation of the finest-LOD mosaic pixel (tile (tx, ty), centre coordinate (rx, ry)); `home` = the atlas tile
// holding that pixel (its previous value is the result where the source has no data)
__device__ __forceinline__ uint32_t rgba8_value_slow(const FusedArgs& A, const RasterDev& r, uint32_t tx, uint32_t rx, uint32_t ty, uint32_t ry,
                                                     uint32_t home_index) {
    const float scale = float(1u << A.lod);
    const uint32_t c = A.m.center_size, b = A.m.border_size, T = A.m.texture_size;
    const Axis ax = split_axis(rx, c, tx, scale, A.tlx, A.brx, r.width);
    const Axis ay = split_axis(ry, c, ty, scale, A.tly, A.bry, r.height);
    const global_u32_t row0 = (global_u32_t)((global_bytes_t)r.data + uint64_t(ay.i0) * r.pitch);
    const global_u32_t row1 = (global_u32_t)((global_bytes_t)r.data + uint64_t(ay.i1) * r.pitch);
    const H4 top = hrow_rgba8(row0[ax.i0], row0[ax.i1], ax.fr), bot = hrow_rgba8(row1[ax.i0], row1[ax.i1], ax.fr);
    if (!(top.valid && bot.valid)) {
        if (home_index == kInvalid || A.prev_zero) return 0;
        return reinterpret_cast<const uint32_t*>(A.atlas)[uint64_t(home_index) * T * T + uint64_t(b + ry) * T + b + rx];
    }
    return vmix_rgba8(top, bot, ay.fr);
}

// split_axis with the two divisions that are exact by construction taken out (same bits): x / 2^lod == x * 2^-lod (no
// underflow: x is 0 or >= 1 / c), and (s - lo) / 1 == s - lo for datasets that span the whole side.
__device__ __forceinline__ Axis split_axis_pow2(uint32_t r, uint32_t c, uint32_t tile, float inv_scale, float lo, float hi, uint32_t dim) {
    const float tc = float(r) / float(c);
    const float s = (float(tile) + tc) * inv_scale;
    const float w = hi - lo;
    const float u = w == 1.0f ? s - lo : (s - lo) / w;
    const float q = u * float(dim) - 0.5f;
    const float fl = floorf(q);
    Axis a;
    a.fr = q - fl;
    const int i = int(fl);
    const int last = int(dim) - 1;
    const int j = i + 1;
    a.i0 = i < 0 ? 0 : (i > last ? last : i);
    a.i1 = j < 0 ? 0 : (j > last ? last : j);
    return a;
}

typedef float direct_f2 __attribute__((ext_vector_type(2)));
struct H4p {  // horizontally blended texel in the 2^8-scaled domain: channels (0, 1) and (2, 3)
    direct_f2 lo, hi;
};
// F(t) for the four channels of a texel (see hrow_rgba8_scaled), two channels per packed operation
__device__ __forceinline__ H4p conv_rgba8_scaled(uint32_t t) {
    const direct_f2 kr = {1.0f / 255.0f, 1.0f / 255.0f};
    const direct_f2 x01 = {float(t & 0xFFu), float((t >> 8) & 0xFFu)}, x23 = {float((t >> 16) & 0xFFu), float(t >> 24)};
    return H4p{__builtin_elementwise_fma(x01, kr, x01), __builtin_elementwise_fma(x23, kr, x23)};
}
__device__ __forceinline__ H4p hrow_rgba8_packed(uint32_t t0, uint32_t t1, float fx) {
    const direct_f2 f2x = {fx, fx}, g2x = {1.0f - fx, 1.0f - fx};
    const H4p a = conv_rgba8_scaled(t0), bq = conv_rgba8_scaled(t1);
    return H4p{a.lo * g2x + bq.lo * f2x, a.hi * g2x + bq.hi * f2x};
}
__device__ __forceinline__ uint32_t vmix_rgba8_packed(const H4p& top, const H4p& bot, float fy) {
    const direct_f2 f2y = {fy, fy}, g2y = {1.0f - fy, 1.0f - fy}, kn = {255.0f / 256.0f, 255.0f / 256.0f}, khalf = {0.5f, 0.5f};
    const direct_f2 lo = khalf + kn * (top.lo * g2y + bot.lo * f2y), hi = khalf + kn * (top.hi * g2y + bot.hi * f2y);
    return uint32_t(lo.x) | (uint32_t(lo.y) << 8) | (uint32_t(hi.x) << 16) | (uint32_t(hi.y) << 24);
}
// the same with the row's (fy, 1 - fy) handed in (vector registers read at a uniform LDS address: no subtraction, no v_readfirstlane)
__device__ __forceinline__ uint32_t vmix_rgba8_weights(const H4p& top, const H4p& bot, float fy, float gy) {
    const direct_f2 f2y = {fy, fy}, g2y = {gy, gy}, kn = {255.0f / 256.0f, 255.0f / 256.0f}, khalf = {0.5f, 0.5f};
    const direct_f2 lo = khalf + kn * (top.lo * g2y + bot.lo * f2y), hi = khalf + kn * (top.hi * g2y + bot.hi * f2y);
    return uint32_t(lo.x) | (uint32_t(lo.y) << 8) | (uint32_t(hi.x) << 16) | (uint32_t(hi.y) << 24);
}
// 2 x 2 average of four texels that all count (rgb != 0), two of the four channels: the ones at bit `sh` (0 or 16) of the
// texels.  Sum order and arithmetic of downsample4_rgba8's common case: ((t00 + t01) + t10) + t11, x 0.25 (exact), quantise
// — 0.5 + (255 / 256) * (sum * 0.25) == 0.5 + (255 / 1024) * sum, both products being the same real number with an exactly
// representable factor.  Returns the two quantised channels in bits 0..15.
__device__ __forceinline__ uint32_t downsample4_rgba8_half(uint32_t t00, uint32_t t01, uint32_t t10, uint32_t t11, uint32_t sh) {
    const direct_f2 kr = {1.0f / 255.0f, 1.0f / 255.0f}, knq = {0.25f * (255.0f / 256.0f), 0.25f * (255.0f / 256.0f)}, khalf = {0.5f, 0.5f};
    auto conv = [&](uint32_t t) -> direct_f2 {
        const uint32_t u = t >> sh;
        const direct_f2 x = {float(u & 0xFFu), float((u >> 8) & 0xFFu)};
        return __builtin_elementwise_fma(x, kr, x);
    };
    const direct_f2 sum = ((conv(t00) + conv(t01)) + conv(t10)) + conv(t11);
    const direct_f2 w = khalf + knq * sum;
    return uint32_t(w.x) | (uint32_t(w.y) << 8);
}
template <int kCtrl>
__device__ __forceinline__ uint32_t quad_dpp(uint32_t v) {
    return uint32_t(__builtin_amdgcn_update_dpp(0, int(v), kCtrl, 0xf, 0xf, true));
}

constexpr uint32_t kDirectRows = 4, kDirectMaxBlocks = 16;

// kRep: a source coarser than the tile grid in y (ratios below 1: the host decides per job) — the chain of a block may stand still (BlockInfo::rep).  The BASELINE
// shapes (4096 over 4064, 8192 over 8128) run the kernel without it: the select it needs costs config 2's albedo job 2 % (same-lease A/B, round 6)
// kSkips: a source finer than the tile grid in y (ratios above 1.02) — an output row's upper source row may lie one past the chain (BlockInfo::skip); up to two such rows of
// a block get that row through two extra pairs requested with the block — by PLAIN loads into ordinary variables: the compiler waits for them in its own (conservative) way.
// The first version issued them from assembly like the chain's and was backed out: a register with an assembly-issued load in flight is not safe from the compiler
// (profiles/r06_gebco_size.txt).
template <bool kRep, bool kSkips = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void fused_direct_rgba8_kernel(FusedArgs A) {
    static_assert(!(kRep && kSkips), "a job's source is coarser or finer than its tile grid, not both");
    constexpr uint32_t kRows = kDirectRows;
    const uint32_t T = A.m.texture_size, b = A.m.border_size, c = A.m.center_size;
    const uint32_t blocks_per_tile = (c + kRows - 1) / kRows;
    // a workgroup runs A.groups consecutive 8-row blocks of one tile: the column axes (one IEEE division per column),
    // neighbour lookups and apron-lane set-up are paid once per sweep instead of once per block
    const uint32_t nb = A.groups, wgs_per_tile = (blocks_per_tile + nb - 1) / nb;
    const uint32_t work = xcd_remap(blockIdx.x, gridDim.x);
    const MainItem it = A.items[work / wgs_per_tile];
    const uint32_t blk_begin = (work % wgs_per_tile) * nb, blk_end = min(blk_begin + nb, blocks_per_tile);
    const uint32_t row_begin = blk_begin * kRows, row_end = min(blk_end * kRows, c);
    const RasterDev raster = A.rasters[it.raster];
    const float scale = float(1u << A.lod), inv_scale = 1.0f / scale;
    const uint32_t tid = threadIdx.x;
    uint32_t* atlas = reinterpret_cast<uint32_t*>(A.atlas);
    const uint32_t tile_texels = T * T;
    uint32_t* tile = atlas + uint64_t(it.atlas_index) * tile_texels;
#ifdef BT_DEBUG_HOOKS
    // (134217728: real-time (100 MHz) stamps per workgroup — entry, set-up done, after each sweep, end — into the atlas's last layer; git history, tools/experiments/direct_probe.py)
    auto stamp = [&](uint32_t slot) {
        if (BT_ABLATE(A, 134217728u) && tid == 0)
            reinterpret_cast<unsigned long long*>(atlas + uint64_t(A.m.atlas_size - 1u) * tile_texels)[work * 8u + slot] = __builtin_amdgcn_s_memrealtime();
    };
#else
    auto stamp = [&](uint32_t) {};
#endif
    stamp(0);
    const uint32_t self4 = A.levels >= 2 ? grid_lookup(A, it.side, A.lod - 1, int(it.x >> 1), int(it.y >> 1)) : kInvalid;
    const uint32_t self3 = A.levels >= 3 ? grid_lookup(A, it.side, A.lod - 2, int(it.x >> 2), int(it.y >> 2)) : kInvalid;

    // per block of the workgroup (+ 2: requests run two blocks ahead): chain = 0 if the block takes the general path, else 0x100 |
    // bit r = (source row r + 1 of the block is one further down than row r); the block's first / last source row
    struct BlockInfo {
        int chain, y_first, y_last;
        uint32_t byte_lo, byte_hi;  // y_first x the raster's pitch
        int rep;                    // bit r (r >= 1): row r uses the same two source rows as row r - 1 (a source coarser than the tiles: round 6)
        int skip;                   // bit r (r >= 1, kSkips): row r's upper source row is the one BEHIND row r - 1's lower one (a source finer than the tiles)
        int pad[1];
    };
    __shared__ Axis s_ay[kDirectMaxBlocks * kRows];
    __shared__ float2 s_wy[kDirectMaxBlocks * kRows];  // (fy, 1 - fy) of the row: read at a uniform address, used from vector registers
    __shared__ BlockInfo s_blk[kDirectMaxBlocks + 2];
    for (uint32_t i = tid; i < row_end - row_begin; i += 256u) {
        const Axis a = split_axis_pow2(row_begin + i, c, it.y, inv_scale, A.tly, A.bry, raster.height);
        s_ay[i] = a;
        s_wy[i] = float2{a.fr, 1.0f - a.fr};
    }
    __syncthreads();
    if (tid < kDirectMaxBlocks + 2) {
        // the fast path rolls over a chain of kRows + 1 source rows: row r's lower source row is row r + 1's upper one, and a pair is
        // one row apart — or the same row, where the source's first / last row is clamped (the tiles along the raster's top and bottom)
        // — or row r repeats row r - 1's pair (a source coarser than the tile grid, ratios below 1: round 6; rounds 2 - 5 sent every such block down the
        // general path, which requests nothing ahead: 0.9 M tiles/s at ratio 0.71 where the chained blocks of ratio 1.008 run at 1.75)
        BlockInfo bi = BlockInfo{0, 0, -1, 0u, 0u, 0, 0, {0}};
        if (blk_begin + tid < blk_end && (blk_begin + tid) * kRows + kRows <= c) {
            const Axis* ay = s_ay + tid * kRows;
            bool ok = true;
            int chain = 0x100, rep = 0, skip = 0, skips = 0;
            for (uint32_t r = 0; ok && r < kRows; r++) {
                const int d = ay[r].i1 - ay[r].i0;
                ok = d == 0 || d == 1;
                if (r == 0 || ay[r].i0 == ay[r - 1].i1) chain |= d << r;  // the next source row of the chain: one further down (or the same, clamped)
                else if (kRep && ay[r].i0 == ay[r - 1].i0 && ay[r].i1 == ay[r - 1].i1) rep |= 1 << r;  // the same pair again: the chain stands still
                else if (kSkips && ay[r].i0 == ay[r - 1].i1 + 1 && skips < 2) {  // one source row is passed over: the chain steps 1 + d, the upper row comes by an extra pair
                    chain |= d << r;
                    skip |= 1 << r;
                    skips++;
                } else ok = false;
            }
            const uint64_t bytes = uint64_t(uint32_t(ay[0].i0)) * raster.pitch;
            if (ok) bi = BlockInfo{chain, ay[0].i0, ay[kRows - 1].i1, uint32_t(bytes), uint32_t(bytes >> 32), rep, skip, {0}};
        }
        s_blk[tid] = bi;
    }
    __syncthreads();
    const global_bytes_t data = (global_bytes_t)raster.data;
    typedef uint8_t __attribute__((address_space(1))) * global_wbytes_t;
    typedef uint32_t __attribute__((address_space(1))) * global_wu32_t;
    const global_wbytes_t tile_bytes = (global_wbytes_t)tile;
    // the parent tiles' quadrants this tile reduces into (their first centre texel)
    const global_wbytes_t base4 = (global_wbytes_t)(atlas + uint64_t(self4 == kInvalid ? 0u : self4) * tile_texels + (b + (it.y & 1u) * (c / 2u)) * T + b + (it.x & 1u) * (c / 2u));
    const global_wbytes_t base3 = (global_wbytes_t)(atlas + uint64_t(self3 == kInvalid ? 0u : self3) * tile_texels + (b + (it.y & 3u) * (c / 4u)) * T + b + (it.x & 3u) * (c / 4u));
    const bool narrow = raster.pitch <= (1ull << 28);  // kRows + 1 rows fit a 32-bit lane offset

    stamp(1);
    const uint32_t c_lanes = (c + 3u) & ~3u;  // whole lane quads take part in the reductions
    // the 2b apron columns of the block's rows ride in spare lanes of the last sweep when there are enough of them (T = 512,
    // b = 2: lanes 252..255): same rows, the column axis of the west / east neighbour (or the own edge column clamped)
    const uint32_t last_cx0 = (c_lanes - 1u) / 256u * 256u;
    const bool aprons_in_sweep = last_cx0 + 256u >= c + 2u * b;
    // A wave's priority falls as it gets on: the arbiter serves the OLDEST wave of a SIMD first, so on a one-generation launch the four
    // waves of a SIMD finish one after the other and the last runs alone at a single wave's issue rate (stamps of
    // git history, tools/experiments/direct_probe.py: workgroups ended 26 .. 62 us after the launch).  Waves that are behind overtake instead.
    const uint32_t prio_step = max(1u, (last_cx0 / 256u + 1u) * (blk_end - blk_begin) / 4u);
    uint32_t prio_left = prio_step, prio_level = 0;
    __builtin_amdgcn_s_setprio(3);
    for (uint32_t cx0 = 0; cx0 < c_lanes || (aprons_in_sweep && cx0 <= last_cx0); cx0 += 256u) {
        const uint32_t cx = cx0 + tid;
        const bool active = cx < c;
        const bool apron_lane = aprons_in_sweep && cx0 == last_cx0 && !active && cx - c < 2u * b;
        const bool used = active || apron_lane;
        uint32_t home = it.atlas_index, home_col = cx, store_px = b + cx;
        Axis ax = Axis{0, 0, 0.0f};
        if (active) {
            ax = split_axis_pow2(cx, c, it.x, inv_scale, A.tlx, A.brx, raster.width);
        } else if (apron_lane) {
            const uint32_t k = cx - c;
            const int rx = k < b ? -1 : 1;
            store_px = k < b ? k : c + k;
            const uint32_t n = grid_lookup(A, it.side, A.lod, int(it.x) + rx, int(it.y));
            if (n != kInvalid) {
                home = n;
                home_col = uint32_t(int(store_px) - int(b) - rx * int(c));
                ax = split_axis_pow2(home_col, c, uint32_t(int(it.x) + rx), inv_scale, A.tlx, A.brx, raster.width);
            } else {
                home_col = k < b ? 0u : c - 1u;
                ax = split_axis_pow2(home_col, c, it.x, inv_scale, A.tlx, A.brx, raster.width);
            }
        }
        const uint32_t half_sh = (tid & 1u) * 16u;  // the channel pair this lane finishes in the lane-split reductions
        const uint32_t off0 = uint32_t(ax.i0) * 4u, off1 = uint32_t(ax.i1) * 4u;  // (host: raster rows shorter than 2^32 bytes)
        // Every instruction a wave issues — scalar ones too — takes a turn of its SIMD's one issue port (git history, tools/experiments/issue_probe.hip): the
        // per-row address steps are lane offsets computed once per sweep instead of scalar additions per row and block.
        uint32_t lo0[kRows + 1], lo1[kRows + 1], so[kRows];
#pragma unroll
        for (uint32_t j = 0; j <= kRows; j++) {
            lo0[j] = off0 + j * uint32_t(raster.pitch);
            lo1[j] = off1 + j * uint32_t(raster.pitch);
            asm volatile("" : "+v"(lo0[j]), "+v"(lo1[j]));
        }
#pragma unroll
        for (uint32_t r = 0; r < kRows; r++) {
            so[r] = (store_px + r * T) * 4u;
            asm volatile("" : "+v"(so[r]));
        }
        uint32_t l4 = (cx >> 1) * 4u, l3 = (cx >> 2) * 4u;
        asm volatile("" : "+v"(l4), "+v"(l3));

        // The blocks of a sweep form a software pipeline over two register sets (the loop is unrolled by two so that no set is ever
        // copied): block k's texels are requested when block k - 2 has consumed its own, in front of that block's stores.
        uint32_t set_a0[kRows + 1], set_a1[kRows + 1], set_b0[kRows + 1], set_b1[kRows + 1];
        uint32_t extra_a[4] = {0u, 0u, 0u, 0u}, extra_b[4] = {0u, 0u, 0u, 0u};  // (kSkips) the upper rows of a block's first / second passing row: texel pairs (i0, i1), plain loads
        BlockInfo info_a, info_b;
        H4p carry_top = H4p{{0.0f, 0.0f}, {0.0f, 0.0f}};
        uint32_t carry_z = 0;
        int carry_row = -1;  // (wave-uniform) the source row carry_top was blended from; -1: none
        auto block_info = [&](uint32_t blk) -> BlockInfo {  // (wave-uniform; blk < blk_begin + kDirectMaxBlocks + 2)
            const BlockInfo* v = s_blk + (blk - blk_begin);
            return BlockInfo{__builtin_amdgcn_readfirstlane(v->chain), __builtin_amdgcn_readfirstlane(v->y_first), __builtin_amdgcn_readfirstlane(v->y_last),
                             uint32_t(__builtin_amdgcn_readfirstlane(int(v->byte_lo))), uint32_t(__builtin_amdgcn_readfirstlane(int(v->byte_hi))),
                             kRep ? __builtin_amdgcn_readfirstlane(v->rep) : 0, kSkips ? __builtin_amdgcn_readfirstlane(v->skip) : 0, {0}};
        };
        // Always 2 x (kRows + 1) loads, whatever the block: the hand-counted waits below rely on it.  A block that takes the general
        // path (chain 0) gets row 0 kRows + 1 times into registers nobody reads.  Issued from assembly and waited for by hand
        // (arrived() below): the compiler's own counted waits assume the fewest operations in flight over all paths of this control
        // flow and end up waiting for the other set and for the stores as well.
        auto request = [&](const BlockInfo& bi, uint32_t (&d0)[kRows + 1], uint32_t (&d1)[kRows + 1], uint32_t (&extra)[4]) {
            global_bytes_t rowp = data + (uint64_t(bi.byte_lo) | uint64_t(bi.byte_hi) << 32);
            global_bytes_t xrow0 = rowp, xrow1 = rowp;
            if (BT_ABLATE(A, 8u)) {  // (8: no source loads)
#pragma unroll
                for (uint32_t j = 0; j <= kRows; j++) {
                    d0[j] = 0x01010101u * (tid + j + 1u) | 1u;
                    d1[j] = 0x01010101u * (tid + j + 2u) | 1u;
                }
            } else if (__builtin_expect(bi.chain == 0x10F && bi.skip == 0 && narrow, 1)) {  // kRows + 1 consecutive rows: one base, the rows in the lane offsets
#pragma unroll
                for (uint32_t j = 0; j <= kRows; j++)
                    asm volatile("global_load_dword %0, %2, %4\n\tglobal_load_dword %1, %3, %4" : "=&v"(d0[j]), "=&v"(d1[j]) : "v"(lo0[j]), "v"(lo1[j]), "s"(rowp));
            } else {
#pragma unroll
                for (uint32_t j = 0; j <= kRows; j++) {
                    asm volatile("global_load_dword %0, %2, %4\n\tglobal_load_dword %1, %3, %4" : "=&v"(d0[j]), "=&v"(d1[j]) : "v"(off0), "v"(off1), "s"(rowp));
                    if (kSkips && j >= 1 && j < kRows && ((uint32_t(bi.skip) >> j) & 1u)) {  // slot j holds row j - 1's lower source row: row j's upper one is the next, its lower one the chain's next
                        rowp += raster.pitch;
                        if ((uint32_t(bi.skip) & ((1u << j) - 1u)) == 0) xrow0 = rowp; else xrow1 = rowp;
                    }
                    rowp += (uint32_t(bi.chain) >> j) & 1u ? raster.pitch : 0u;
                }
            }
            if constexpr (kSkips) {
                if (bi.skip != 0) {  // (wave-uniform) plain loads: ordinary variables, the compiler's own waits
                    extra[0] = *(global_u32_t)(xrow0 + off0);
                    extra[1] = *(global_u32_t)(xrow0 + off1);
                    extra[2] = *(global_u32_t)(xrow1 + off0);
                    extra[3] = *(global_u32_t)(xrow1 + off1);
                }
            }
        };
        // row j of a set has arrived when at most the set's later rows and the other set's request (always issued after it: kRows + 1
        // pairs) are in flight; whatever else was issued in between only makes the count conservative
        auto arrived = [&](uint32_t j, uint32_t& t0, uint32_t& t1) {
            switch (j) {
                case 0: asm volatile("s_waitcnt vmcnt(18)" : "+v"(t0), "+v"(t1)); break;
                case 1: asm volatile("s_waitcnt vmcnt(16)" : "+v"(t0), "+v"(t1)); break;
                case 2: asm volatile("s_waitcnt vmcnt(14)" : "+v"(t0), "+v"(t1)); break;
                case 3: asm volatile("s_waitcnt vmcnt(12)" : "+v"(t0), "+v"(t1)); break;
                default: asm volatile("s_waitcnt vmcnt(10)" : "+v"(t0), "+v"(t1)); break;
            }
        };
        static_assert(kRows == 4, "arrived() counts 2 x (kRows + 1) loads per request");
        auto block = [&](uint32_t blk, BlockInfo& bi, uint32_t (&raw0)[kRows + 1], uint32_t (&raw1)[kRows + 1], uint32_t (&extra)[4]) {
            if (--prio_left == 0) {
                prio_left = prio_step;
                prio_level++;
                if (prio_level == 1u) __builtin_amdgcn_s_setprio(2);
                else if (prio_level == 2u) __builtin_amdgcn_s_setprio(1);
                else __builtin_amdgcn_s_setprio(0);
            }
            const uint32_t cr0 = blk * kRows, nrows = min(kRows, c - cr0);
            const Axis* ay_blk = s_ay + (blk - blk_begin) * kRows;
            uint32_t out[kRows];
#pragma unroll
            for (uint32_t r = 0; r < kRows; r++) out[r] = 0;
            // ---- the fast path: the block's rows use a chain of kRows + 1 source rows (requested two blocks ago), and no texel any
            // lane of the wave reads is "no data" (channel 0 == 0, split.wgsl:34): no per-pixel validity, plain stores
            const bool chained = bi.chain != 0;
            bool fast = chained;
            uint32_t z = 1;
            if (__builtin_expect(chained, 1)) {
                // rolling over the source rows as they arrive; the no-data test rides along and is evaluated before anything is stored.
                // The block's first source row is usually the previous block's last: its blended texel (and its no-data byte) is carried
                const float2* wy = s_wy + (blk - blk_begin) * kRows;
                H4p top;
                if (__builtin_expect(carry_row == bi.y_first, 1)) {
                    z = carry_z;
                    top = carry_top;
                } else {
                    arrived(0, raw0[0], raw1[0]);
                    z = min(raw0[0] & 0xFFu, raw1[0] & 0xFFu);
                    top = hrow_rgba8_packed(raw0[0], raw1[0], ax.fr);
                }
                if constexpr (!kRep) {
                    const uint32_t x_first0 = extra[0], x_first1 = extra[1], x_second0 = extra[2], x_second1 = extra[3];  // (kSkips: requested with this block's chain)
#pragma unroll
                    for (uint32_t r = 0; r < kRows; r++) {
                        arrived(r + 1, raw0[r + 1], raw1[r + 1]);
                        const uint32_t z_row = min(raw0[r + 1] & 0xFFu, raw1[r + 1] & 0xFFu);
                        z = min(z, z_row);
                        const H4p bot = hrow_rgba8_packed(raw0[r + 1], raw1[r + 1], ax.fr);
                        if constexpr (kSkips) {
                            if (r > 0 && ((uint32_t(bi.skip) >> r) & 1u) != 0) {  // (wave-uniform) the upper row is the one behind the chain's: the block's first or second extra pair
                                const bool second = (uint32_t(bi.skip) & ((1u << r) - 1u)) != 0;
                                const uint32_t x0 = second ? x_second0 : x_first0, x1 = second ? x_second1 : x_first1;
                                z = min(z, min(x0 & 0xFFu, x1 & 0xFFu));
                                top = hrow_rgba8_packed(x0, x1, ax.fr);
                            }
                        }
                        const float2 w = wy[r];
                        out[r] = vmix_rgba8_weights(top, bot, w.x, w.y);
                        top = bot;
                        if (r + 1 == kRows) carry_z = z_row;
                    }
                } else {
                    H4p held = top;      // the upper row of the previous output row (a repeated pair keeps it)
                    uint32_t z_row = 1;
#pragma unroll
                    for (uint32_t r = 0; r < kRows; r++) {
                        arrived(r + 1, raw0[r + 1], raw1[r + 1]);
                        const bool again = r > 0 && ((uint32_t(bi.rep) >> r) & 1u) != 0;  // (wave-uniform) the same pair as the row above: nothing new to blend
                        H4p bot = top;
                        if (!again) {
                            z_row = min(raw0[r + 1] & 0xFFu, raw1[r + 1] & 0xFFu);
                            z = min(z, z_row);
                            bot = hrow_rgba8_packed(raw0[r + 1], raw1[r + 1], ax.fr);
                        }
                        const H4p upper = again ? held : top;
                        const float2 w = wy[r];
                        out[r] = vmix_rgba8_weights(upper, bot, w.x, w.y);
                        held = upper;
                        top = bot;
                        if (r + 1 == kRows) carry_z = z_row;
                    }
                }
                carry_top = top;
                carry_row = bi.y_last;
            } else {
                carry_row = -1;
            }
            // the block after the next one, into the registers just consumed — BEFORE this block's stores: the memory counter is in
            // order, so a wait for a request covers everything issued before it.  Requests issued behind the previous block's stores (as
            // a plain prefetch would) make every block wait for a store acknowledge.
            bi = block_info(blk + 2u);
            request(bi, raw0, raw1, extra);
            if (__builtin_expect(chained, 1)) {
                if (__builtin_expect(__ballot(used && z == 0u) != 0ull, 0)) {
                    fast = false;  // (wave-uniform) the general path below redoes the block
                } else if (used && !BT_ABLATE(A, 2u)) {  // (2: no finest stores)
                    const global_wbytes_t rowq = tile_bytes + (b + cr0) * T * 4u;  // uniform row pointer + 32-bit lane offsets, like the loads
#pragma unroll
                    for (uint32_t r = 0; r < kRows; r++)  // (written out: the compiler widened the lane offsets to 64-bit vector additions)
                        asm volatile("global_store_dword %0, %1, %2" ::"v"(so[r]), "v"(out[r]), "s"(rowq) : "memory");
                }
            }
            if (__builtin_expect(!fast, 0)) {
                // ---- the general path, row by row: a pixel without data is not stored (it keeps the atlas value,
                // split.wgsl:37-42) and its previous value is fetched for the reductions; an apron pixel copies what its
                // home tile holds
                bool keep[kRows];
#pragma unroll
                for (uint32_t r = 0; r < kRows; r++) keep[r] = false;
#pragma unroll
                for (uint32_t r = 0; r < kRows; r++) {
                    if (r >= nrows) continue;
                    const Axis ay = ay_blk[r];
                    const global_u32_t row0 = (global_u32_t)(data + uint64_t(ay.i0) * raster.pitch), row1 = (global_u32_t)(data + uint64_t(ay.i1) * raster.pitch);
                    const H4 top = hrow_rgba8(row0[ax.i0], row0[ax.i1], ax.fr), bot = hrow_rgba8(row1[ax.i0], row1[ax.i1], ax.fr);
                    keep[r] = !(top.valid && bot.valid);
                    out[r] = vmix_rgba8(top, bot, ay.fr);
                }
                bool any_keep = false;
#pragma unroll
                for (uint32_t r = 0; r < kRows; r++) {
                    if (used && r < nrows && !keep[r]) tile[(b + cr0 + r) * T + store_px] = out[r];
                    any_keep = any_keep || keep[r];
                }
                if (__ballot(used && any_keep)) {
#pragma unroll
                    for (uint32_t r = 0; r < kRows; r++)
                        if (used && r < nrows && keep[r]) {
                            // (A.prev_zero: a fresh atlas — the kept value is bt_atlas_create's 0, and the pixel, unwritten, already holds it)
                            out[r] = A.prev_zero ? 0u : atlas[uint64_t(home) * tile_texels + (b + cr0 + r) * T + b + home_col];
                            if (apron_lane && !A.prev_zero) tile[(b + cr0 + r) * T + store_px] = out[r];
                        }
                }
                // the fetched values arrive HERE: left pending, the compiler would guard the reductions below — which the fast path
                // shares — with a wait for every memory operation in flight, the next blocks' requests included
#pragma unroll
                for (uint32_t r = 0; r < kRows; r++) asm volatile("" : "+v"(out[r]));
            }
            if (A.levels < 2 || self4 == kInvalid || BT_ABLATE(A, 1u)) return;  // (1: no pyramid)
            // ---- LOD-1: rows (2i, 2i+1) in registers, columns (cx, cx + 1) in the lane pair.  Common case (every texel of the
            // wave counts, rgb != 0): the two lanes of a pair split the four channels — each gathers the pair's four texels
            // (left column = even lane, downsample.wgsl OFFSETS order) and finishes two channels; the halves meet by one DPP swap.
            uint32_t q[kRows / 2];
            uint32_t zr = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t r = 0; r < kRows; r++) zr = min(zr, out[r] << 8);
            if (__builtin_expect(nrows == kRows && !__ballot(active && zr == 0u), 1)) {
#pragma unroll
                for (uint32_t i = 0; i < kRows / 2; i++) {
                    const uint32_t l0 = quad_dpp<0xA0>(out[2 * i]), l1 = quad_dpp<0xA0>(out[2 * i + 1]);  // quad_perm [0, 0, 2, 2]: the even lane's column
                    const uint32_t r0 = quad_dpp<0xF5>(out[2 * i]), r1 = quad_dpp<0xF5>(out[2 * i + 1]);  // quad_perm [1, 1, 3, 3]: the odd lane's
                    const uint32_t mine = downsample4_rgba8_half(l0, l1, r0, r1, half_sh) << half_sh;
                    q[i] = mine | quad_dpp<0xB1>(mine);
                }
            } else {
#pragma unroll
                for (uint32_t i = 0; i < kRows / 2; i++) {
                    // (the odd lane's own result sums in another order; it is replaced by the even lane's, the one the reference order gives)
                    const uint32_t v = downsample4_rgba8(out[2 * i], out[2 * i + 1], quad_dpp<0xB1>(out[2 * i]), quad_dpp<0xB1>(out[2 * i + 1]));
                    q[i] = quad_dpp<0xA0>(v);
                }
            }
            {   // centre texels; the parents' aprons come from the tail launch
                const global_wbytes_t row4 = base4 + (cr0 >> 1) * T * 4u;
                if (active && (tid & 1u) == 0) {
#pragma unroll
                    for (uint32_t i = 0; i < kRows / 2; i++)
                        if (2 * i + 1 < nrows) *(global_wu32_t)(row4 + i * T * 4u + l4) = q[i];
                }
            }
            if (A.levels < 3 || self3 == kInvalid) return;
            // ---- LOD-2: every lane of a quad holds the quad's two LOD-1 texels of a row (lanes 0, 1 the left, 2, 3 the right)
            uint32_t w[kRows / 4];
            uint32_t zq = 0xFFFFFFFFu;
#pragma unroll
            for (uint32_t i = 0; i < kRows / 2; i++) zq = min(zq, q[i] << 8);
            if (__builtin_expect(nrows == kRows && !__ballot(active && zq == 0u), 1)) {
#pragma unroll
                for (uint32_t j = 0; j < kRows / 4; j++) {
                    const uint32_t l0 = quad_dpp<0x00>(q[2 * j]), l1 = quad_dpp<0x00>(q[2 * j + 1]);  // quad_perm [0, 0, 0, 0]
                    const uint32_t r0 = quad_dpp<0xAA>(q[2 * j]), r1 = quad_dpp<0xAA>(q[2 * j + 1]);  // quad_perm [2, 2, 2, 2]
                    const uint32_t mine = downsample4_rgba8_half(l0, l1, r0, r1, half_sh) << half_sh;
                    w[j] = mine | quad_dpp<0xB1>(mine);
                }
            } else {
#pragma unroll
                for (uint32_t j = 0; j < kRows / 4; j++)
                    w[j] = downsample4_rgba8(quad_dpp<0x00>(q[2 * j]), quad_dpp<0x00>(q[2 * j + 1]), quad_dpp<0xAA>(q[2 * j]), quad_dpp<0xAA>(q[2 * j + 1]));
            }
            {
                const global_wbytes_t row3 = base3 + (cr0 >> 2) * T * 4u;
#pragma unroll
                for (uint32_t j = 0; j < kRows / 4; j++)
                    if (active && (tid & 3u) == 0 && 4 * j + 3 < nrows) *(global_wu32_t)(row3 + j * T * 4u + l3) = w[j];
            }
        };
        if (BT_ABLATE(A, 16u)) continue;  // (16: set-up and aprons only — timing experiment)
        info_a = block_info(blk_begin);
        request(info_a, set_a0, set_a1, extra_a);
        info_b = block_info(blk_begin + 1u);
        request(info_b, set_b0, set_b1, extra_b);
        for (uint32_t blk = blk_begin;;) {  // (no path from one use of a set to its next use without the other set's block in between: the counted waits rely on it)
            block(blk, info_a, set_a0, set_a1, extra_a);
            if (++blk >= blk_end) break;
            block(blk, info_b, set_b0, set_b1, extra_b);
            if (++blk >= blk_end) break;
        }
        stamp(2u + cx0 / 256u);
    }

    // the tile's own apron: stitch.wgsl:53-118 with the neighbour's centre pixel evaluated from the source (its own
    // formula), or the own centre clamped where the neighbour does not exist (same-face neighbours only: cube seams
    // are re-stitched by the batched kernel afterwards)
    // The 2b whole apron rows (above the first block, below the last) are shared out over ALL workgroups of the tile, an equal
    // run of pixels each: left to the first / last workgroup they were a 10 us tail on one resident generation of workgroups.
    const uint32_t n_cols = aprons_in_sweep ? 0u : (row_end - row_begin) * 2u * b;
    const uint32_t row_px = 2u * b * T, share = (row_px + wgs_per_tile - 1u) / wgs_per_tile;
    const uint32_t px_begin = min((work % wgs_per_tile) * share, row_px), px_end = min(px_begin + share, row_px);
    for (uint32_t i = tid; i < n_cols + (px_end - px_begin) && !BT_ABLATE(A, 4u); i += 256u) {  // (4: no apron rows)
        uint32_t px, py;
        if (i < n_cols) {  // the apron columns of this workgroup's rows
            const uint32_t r = i / (2u * b), k = i % (2u * b);
            px = k < b ? k : c + k;
            py = b + row_begin + r;
        } else {  // apron rows: b above the centre, then b below
            const uint32_t e = px_begin + (i - n_cols), r = e / T;
            px = e % T;
            py = r < b ? r : c + r;
        }
        const int rx = px < b ? -1 : (px >= b + c ? 1 : 0), ry = py < b ? -1 : (py >= b + c ? 1 : 0);
        const uint32_t n = grid_lookup(A, it.side, A.lod, int(it.x) + rx, int(it.y) + ry);
        const bool have = n != kInvalid;
        const uint32_t sx = have ? uint32_t(int(it.x) + rx) : it.x, sy = have ? uint32_t(int(it.y) + ry) : it.y;
        const uint32_t qx = have ? uint32_t(int(px) - int(b) - rx * int(c)) : min(max(px, b), b + c - 1u) - b;
        const uint32_t qy = have ? uint32_t(int(py) - int(b) - ry * int(c)) : min(max(py, b), b + c - 1u) - b;
        tile[py * T + px] = rgba8_value_slow(A, raster, sx, qx, sy, qy, have ? n : it.atlas_index);
    }
    stamp(4);
}

// exhaustive device check of the fast unorm conversion against correctly rounded division
__global__ void selftest_kernel(uint32_t* failures) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 65536u) {
        volatile float d = 65535.0f;
        if (unorm16_to_float(t) != float(t) / d) atomicAdd(failures, 1u);
        // the fast loop's scaled conversion: fma(x, r, x) == 65536 * RN(t / 65535)
        const float x = float(t);
        if (__builtin_fmaf(x, 1.0f / 65535.0f, x) != 65536.0f * (x / d)) atomicAdd(failures, 1u);
    }
    if (t < 256u) {  // the 8-bit conversion of the batched kernels (same construction)
        volatile float d = 255.0f;
        const float x = float(t), r = 1.0f / 255.0f, q0 = x * r;
        if (__builtin_fmaf(__builtin_fmaf(-q0, 255.0f, x), r, q0) != x / d) atomicAdd(failures, 1u);
        if (__builtin_fmaf(x, r, x) != 256.0f * (x / d)) atomicAdd(failures, 1u);  // fused_direct's scaled conversion
    }
}

}  // namespace

// =============================================================================== host side: planning

struct FusedJobDev {  // one fused launch of a compiled queue
    FusedArgs args;
    uint32_t attachment;
    uint32_t lds_pad = 0;    // profiling build only (BT_FUSED_LDS_PAD at plan time): extra dynamic LDS per workgroup
    bool dma = false;        // fused_main stages through LDS-DMA (every raster of the job 16-byte aligned in base and pitch)
    bool dma_only = false;   // ... and only so: the window has more 16-byte pieces than the register staging batches (run-time-pitch DMA variant)
    bool direct_rep = false;  // fused_direct: the source is coarser than the tile grid in y (ratio below 1): the variant whose chains may stand still
    bool direct_skips = false;  // ... finer (ratio above 1.02): the variant whose chains may pass over a source row
    std::vector<MainItem> host_items;  // fused_main's / fused_direct's items as uploaded (tile-row order): streamed runs cut fused_main's into bands, fused_source_window reads both
    bool direct = false;     // a fused_direct launch (reads the source texel by texel: no staged window)
    uint32_t seam_first = 0;  // fused_tail with seam workgroups: its tasks are p->tasks_dev[seam_first ...] (args.seam_count of them)
    float tly = 0.0f, bry = 1.0f;
    // fused_main / fused_direct: the atlas layers of the job's finest tiles (a no-data pixel's "previous value" is read from one of them) and of
    // all its tiles (written by this job's launches); fused_begin_run decides args.prev_zero from Attachment::written before every run
    std::vector<uint32_t> finest_layers, all_layers;
};

// the fused path's per-queue state, owned by the bt_preprocessor that compiled it (bt_preprocessor::fused)
struct FusedState {
    std::vector<FusedJobDev> jobs;
    std::vector<void*> allocs;  // device buffers of the jobs (grids, item lists)
    std::vector<uint8_t> whole_raster;  // [raster]: a launch without an item list (the hybrid plan's batched split) reads it: no window is known
};

}  // namespace bt

#include <algorithm>
namespace bt {

static FusedState& state_of(bt_preprocessor* p) {
    if (!p->fused) p->fused = new FusedState();
    return *p->fused;
}

void fused_release(bt_preprocessor* p) {
    if (!p->fused) return;
    for (void* d : p->fused->allocs) hipFree(d);
    delete p->fused;
    p->fused = nullptr;
}

template <typename V>
static bt_status upload_vector(bt_preprocessor* p, const std::vector<V>& v, const V** out) {
    void* d = nullptr;
    BT_HIP(hipMalloc(&d, v.size() * sizeof(V) ? v.size() * sizeof(V) : 1));
    state_of(p).allocs.push_back(d);
    if (!v.empty()) BT_HIP(hipMemcpy(d, v.data(), v.size() * sizeof(V), hipMemcpyHostToDevice));
    *out = (const V*)d;
    return BT_OK;
}

// A job (one preprocess_tile / preprocess_spherical call) qualifies for the fused path when
//  - the attachment is R16 with T <= 512, even b, c % 4 == 0, c >= 2b;
//  - at every LOD but the finest, each queued tile has all four children queued (full quadtree below it).
// Otherwise the whole queue runs on the generic kernels.
bool fused_plan(bt_preprocessor* p, bt_atlas* a, std::vector<TaskDev>& tasks, std::vector<Launch>& plan) {
    FusedState& state = state_of(p);
    for (void* d : state.allocs) hipFree(d);
    state.allocs.clear();
    std::vector<FusedJobDev>& jobs = state.jobs;
    jobs.clear();
    state.whole_raster.assign(p->rasters.size(), 0);
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
        // fused_main: R16, T <= 512, even b <= 8.  Otherwise (Rgba8, large tiles, odd borders) the HYBRID plan: the batched
        // split + stitch kernels produce the finest LOD, fused_tail (format-generic) everything below it, three LODs per
        // launch, aprons pushed — instead of one downsample launch per LOD and a stitch over every tile.
        const bool tail_ok = (m.format == BT_FORMAT_R16 || m.format == BT_FORMAT_RGBA8) && (m.center_size & 3u) == 0 &&
                             m.center_size >= 2 * m.border_size && m.border_size != 0 && (m.format != BT_FORMAT_R16 || (m.border_size & 1u) == 0);
        const bool main_ok = tail_ok && m.format == BT_FORMAT_R16 && m.texture_size <= 512 && m.border_size <= 8;
        if (!tail_ok) return false;
        // Rgba8: fused_direct (no LDS staging) produces the finest LOD with its aprons and the two parent LODs
        const bool direct = !main_ok && m.format == BT_FORMAT_RGBA8;
        const bool hybrid = !main_ok && !direct;
        // (layer x tile texels is formed in 64 bits everywhere: an attachment of 2^32 texels or more — 16384 tiles of 512^2 — takes the fused plans
        // like any other; rounds 2 - 6 sent such atlases to the batched kernels, found with a GEBCO-sized job at the end of round 6)
        const uint32_t lod_hi = splits[0]->coord.lod;
        uint32_t lod_lo = lod_hi;
        for (const Task* t : downs) lod_lo = std::min(lod_lo, t->coord.lod);
        if (lod_hi > 13) return false;  // dense per-LOD grids (4^lod entries) and 32-bit grid offsets
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
        auto in_face = [](const bt_tile_coordinate& c) { return c.x < (1u << c.lod) && c.y < (1u << c.lod); };
        for (const Task* t : splits) {
            if (t->coord.side >= sides || t->coord.lod != lod_hi || !in_face(t->coord)) return false;
            cell(t->coord) = t->atlas_index;
        }
        for (const Task* t : downs) {
            if (t->coord.side >= sides || t->coord.lod < lod_lo || t->coord.lod > lod_hi || !in_face(t->coord)) return false;
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
        // The fused kernels take a tile's apron from the job's own grid; the queue recorded the neighbours the ATLAS holds
        // (stitch_and_save_layer -> get_tile).  They differ when tiles of an earlier job or dataset border this one: then
        // only the generic path stitches across that seam, so the job does not qualify.
        for (const Task* t : stitches) {
            if (t->coord.side >= sides || t->coord.lod < lod_lo || t->coord.lod > lod_hi || !in_face(t->coord)) return false;
            if (cell(t->coord) != t->atlas_index) return false;
            bt_tile_coordinate nb[8];
            tile_neighbours(t->coord, false, nb);  // same-face neighbours (cube seams are re-stitched by the generic kernel)
            for (int i = 0; i < 8; i++) {
                const bool same_face = !is_invalid(nb[i]);
                const uint32_t in_grid = same_face ? cell(nb[i]) : kInvalid;
                const bool recorded_same_face = t->rel[i].atlas_index != BT_INVALID_ATLAS_INDEX && t->rel[i].coordinate.side == t->coord.side && same_face;
                if (same_face && in_grid != (recorded_same_face ? t->rel[i].atlas_index : kInvalid)) return false;
            }
        }
        // all split tasks share the dataset rectangle
        for (const Task* t : splits)
            if (t->tl[0] != splits[0]->tl[0] || t->tl[1] != splits[0]->tl[1] || t->br[0] != splits[0]->br[0] || t->br[1] != splits[0]->br[1])
                return false;

        // ---- sharding (multi-GPU).  Unit = one column strip of one side at the granularity of the coarsest LOD the main
        // kernel produces (a strip = 2^(levels-1) finest columns = one column of that LOD), units numbered side-major;
        // rank r owns the units [r * U / world, (r + 1) * U / world).  A unit's tiles of a LOD are contiguous atlas layers
        // (x-major allocation order), so the exchange is a list of contiguous layer runs, each with its owning rank.
        const uint32_t world = p->shard_world, rank = p->shard_rank;
        const uint32_t nlods_all = lod_hi - lod_lo + 1, main_levels_all = std::min(3u, nlods_all);
        const uint32_t strips = 1u << (lod_hi - (main_levels_all - 1)), units = sides * strips;
        bool shard = world > 1 && units % world == 0 && !hybrid;  // (fused_direct shards like fused_main: finest tiles complete from the source, parent centres inside a strip, everything else after the exchange)
        std::vector<bt_shard_range> ranges;
        std::vector<bt_shard_piece> pieces;
        if (shard) {
            const uint32_t units_per_rank = units / world;
            for (uint32_t side = 0; side < sides && shard; side++)
                for (uint32_t k = 0; k < main_levels_all && shard; k++) {
                    const uint32_t lod = lod_hi - k, n = 1u << lod;
                    const uint32_t off = grid_offsets[side * 32 + lod];
                    const uint32_t base = grids[off];
                    for (size_t i = 0; i < size_t(n) * n; i++)
                        if (base == kInvalid || grids[off + i] != base + i) shard = false;  // not the fresh x-major layout
                    const uint32_t cols_per_strip = n / strips;
                    // maximal runs of strips with one owner
                    for (uint32_t strip = 0; strip < strips;) {
                        const uint32_t owner = (side * strips + strip) / units_per_rank;
                        uint32_t end = strip + 1;
                        while (end < strips && (side * strips + end) / units_per_rank == owner) end++;
                        pieces.push_back({ai, side, lod, base + strip * cols_per_strip * n, (end - strip) * cols_per_strip * n, owner});
                        strip = end;
                    }
                    // the regular case (one side, every rank an equal run): also expressible as ONE in-place all-gather
                    if (sides == 1) ranges.push_back({ai, side, lod, base, n / world * n});
                }
        }
        if (world > 1 && !shard) {
            ranges.clear();
            pieces.clear();
        }

        std::vector<MainItem> items;
        for (const Task* t : splits) {
            if (shard) {
                const uint32_t unit = t->coord.side * strips + (t->coord.x >> (main_levels_all - 1));
                if (unit / (units / world) != rank) continue;
            }
            items.push_back({t->coord.side, t->coord.x, t->coord.y, t->atlas_index, uint32_t(t->raster)});
        }
        // Workgroup order = tile rows (y outer, x inner) instead of the queue's x-major order: an XCD then streams whole
        // source rows (its 128 concurrent workgroups cover 4 tile rows x all columns), and x neighbours run on the same XCD
        // at the same time, so the apron bytes one pushes into the other's parent rows merge in one L2.  16k job: 333 -> 285 us.
#ifdef BT_DEBUG_HOOKS
        if (!getenv("BT_FUSED_XMAJOR"))
#endif
            std::stable_sort(items.begin(), items.end(), [](const MainItem& a, const MainItem& b2) {
                return a.side != b2.side ? a.side < b2.side : (a.y != b2.y ? a.y < b2.y : a.x < b2.x);
            });
#ifdef BT_DEBUG_HOOKS
        // workgroup -> tile experiments (git history, tools/experiments/order_search.py): a file of item_count u32, position i of the (XCD-contiguous)
        // work order runs the tile at that position of the tile-row order
        if (const char* e = getenv("BT_FUSED_ORDER")) {
            std::vector<uint32_t> perm(items.size());
            FILE* f = fopen(e, "rb");
            const bool ok = f && fread(perm.data(), 4, perm.size(), f) == perm.size();
            if (f) fclose(f);
            if (ok) {
                std::vector<MainItem> sorted = items;
                std::vector<uint8_t> seen(items.size(), 0);
                bool valid = true;
                for (uint32_t v : perm) valid = valid && v < items.size() && !seen[v] && (seen[v] = 1);
                if (valid)
                    for (size_t i = 0; i < items.size(); i++) items[i] = sorted[perm[i]];
                else
                    fprintf(stderr, "BT_FUSED_ORDER: %s is not a permutation of %zu items, ignored\n", e, items.size());
            }
        }
#endif
        if (shard) {
            p->shard_ranges.insert(p->shard_ranges.end(), ranges.begin(), ranges.end());
            p->shard_pieces.insert(p->shard_pieces.end(), pieces.begin(), pieces.end());
        }

        FusedArgs args{};
        args.m = m;
#ifdef BT_DEBUG_HOOKS
        if (const char* e = getenv("BT_FUSED_ABLATE")) args.ablate = uint32_t(strtoul(e, nullptr, 0));
#endif
        args.atlas = (uint16_t*)at.level0;
        args.rasters = p->rasters_dev;  // (re)allocated by bt_preprocessor_run before the first launch
        args.tlx = splits[0]->tl[0];
        args.tly = splits[0]->tl[1];
        args.brx = splits[0]->br[0];
        args.bry = splits[0]->br[1];
        args.sides = sides;
        args.grid_lod_lo = lod_lo;
        args.grid_lod_hi = lod_hi;
        args.grid_sides = sides;
        {   // the closed form of grid_lookup: does every entry follow it?
            const uint32_t hi4 = 4u << (2u * lod_hi), per_side = (hi4 - (1u << (2u * lod_lo))) / 3u, first = grids.empty() ? 0u : grids[grid_offsets[lod_hi]];
            bool regular = first != kInvalid;
            for (uint32_t side = 0; side < sides && regular; side++)
                for (uint32_t lod = lod_lo; lod <= lod_hi && regular; lod++) {
                    const uint32_t off = grid_offsets[side * 32 + lod], base = first + side * per_side + (hi4 - (4u << (2u * lod))) / 3u;
                    for (size_t i = 0; i < (size_t(1) << (2 * lod)); i++)
                        if (grids[off + i] != base + uint32_t(i)) { regular = false; break; }
                }
            args.regular = regular ? 1u : 0u;
            args.reg_first = first;
#ifdef BT_DEBUG_HOOKS
            if (getenv("BT_FUSED_NO_REGULAR")) args.regular = 0;
#endif
        }
        if (upload_vector(p, items, &args.items) || upload_vector(p, grids, &args.grids))
            return false;

        const uint64_t bpp = m.pixel_size, Tt = m.texture_size, cc = m.center_size;
        uint64_t source_bytes = 0;
        {
            std::vector<bool> seen(p->rasters.size(), false);
            for (const Task* t : splits)
                if (!seen[t->raster]) {
                    seen[t->raster] = true;
                    source_bytes += uint64_t(p->rasters[t->raster].dev.width) * p->rasters[t->raster].dev.height * bpp;
                }
        }
        std::vector<uint32_t> finest_layers, all_layers;  // (Attachment::written bookkeeping: fused_begin_run)
        for (uint32_t side = 0; side < sides; side++)
            for (uint32_t lod = lod_lo; lod <= lod_hi; lod++) {
                const uint32_t off = grid_offsets[side * 32 + lod];
                for (size_t i = 0; i < (size_t(1) << (2 * lod)); i++)
                    if (grids[off + i] != kInvalid) {
                        all_layers.push_back(grids[off + i]);
                        if (lod == lod_hi) finest_layers.push_back(grids[off + i]);
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
        const uint32_t main_levels = hybrid ? 1u : std::min(3u, nlods);
        auto device_task = [](const Task& t) {
            TaskDev d{};
            d.atlas_index = t.atlas_index;
            d.side = t.coord.side;
            d.lod = t.coord.lod;
            d.x = t.coord.x;
            d.y = t.coord.y;
            d.tlx = t.tl[0];
            d.tly = t.tl[1];
            d.brx = t.br[0];
            d.bry = t.br[1];
            d.raster = t.raster < 0 ? 0u : uint32_t(t.raster);
            for (int i = 0; i < 8; i++) {
                d.rel_index[i] = t.rel[i].atlas_index;
                d.rel_side[i] = t.rel[i].coordinate.side;
            }
            return d;
        };
        if (hybrid) {
            Launch ls{};
            ls.kind = kLaunchSplit;
            ls.attachment = ai;
            ls.first_task = uint32_t(tasks.size());
            for (const Task* t : splits) {
                tasks.push_back(device_task(*t));
                if (t->raster >= 0 && size_t(t->raster) < state.whole_raster.size()) state.whole_raster[size_t(t->raster)] = 1;  // (every rank splits every tile)
            }
            ls.task_count = uint32_t(splits.size());
            ls.algorithmic_bytes = source_bytes + uint64_t(splits.size()) * Tt * Tt * bpp;
            plan.push_back(ls);
            Launch lt{};
            lt.kind = kLaunchStitch;
            lt.attachment = ai;
            lt.first_task = uint32_t(tasks.size());
            for (const Task* t : stitches)
                if (t->coord.lod == lod_hi) tasks.push_back(device_task(*t));
            lt.task_count = uint32_t(tasks.size()) - lt.first_task;
            lt.algorithmic_bytes = uint64_t(lt.task_count) * 2 * (2 * m.border_size * (Tt + cc)) * bpp;
            if (lt.task_count) plan.push_back(lt);
        } else if (direct) {
            FusedJobDev job{args, ai};
            job.args.lod = lod_hi;
            job.args.levels = main_levels;
            job.args.item_count = uint32_t(items.size());
            job.host_items = items;
            job.direct = true;
            job.finest_layers = finest_layers;
            job.all_layers = all_layers;
            job.tly = args.tly;
            job.bry = args.bry;
            {   // source rows per tile row: below 1 output rows repeat source-row pairs
                const double mosaic = double(m.center_size) * double(1u << lod_hi);
                for (const Task* t : splits) {
                    const RasterDev& r = p->rasters[t->raster].dev;
                    const double ratio = double(r.height) / (double(args.bry - args.tly) * mosaic);
                    if (ratio < 0.9999) job.direct_rep = true;
                    if (ratio > 1.02) job.direct_skips = true;  // (the BASELINE shapes, 1.008, pass over a row in one block of 32: they stay on the plain kernel)
                }
            }
            {   // row blocks per workgroup: as many as keep at least one resident generation (1024 workgroups) busy
                const uint64_t blocks = uint64_t(items.size()) * ((m.center_size + kDirectRows - 1) / kDirectRows);
                job.args.groups = uint32_t(std::min<uint64_t>(kDirectMaxBlocks, std::max<uint64_t>(1, (blocks + 1023) / 1024)));
#ifdef BT_DEBUG_HOOKS
                if (const char* e = getenv("BT_FUSED_PARTS")) job.args.groups = std::max(1u, std::min(kDirectMaxBlocks, uint32_t(atoi(e))));
#endif
            }
            Launch ld{};
            ld.kind = kLaunchFusedDirect;
            ld.attachment = ai;
            ld.task_count = uint32_t(items.size());
            ld.aux0 = uint32_t(jobs.size());
            ld.algorithmic_bytes = source_bytes;
            for (uint32_t k = 0; k < main_levels; k++) ld.algorithmic_bytes += tiles_at(lod_hi - k) * Tt * Tt * bpp;
            jobs.push_back(job);
            plan.push_back(ld);
        } else {
        FusedJobDev main_job{args, ai};
#ifdef BT_DEBUG_HOOKS
        if (const char* e = getenv("BT_FUSED_LDS_PAD")) main_job.lds_pad = uint32_t(atoi(e));
#endif
        main_job.args.lod = lod_hi;
        main_job.args.levels = main_levels;
        main_job.args.item_count = uint32_t(items.size());
        main_job.host_items = items;
        main_job.finest_layers = finest_layers;
        main_job.all_layers = all_layers;
        main_job.tly = args.tly;
        main_job.bry = args.bry;
        {
            const uint32_t chunks = (m.center_size + kMainRows - 1) / kMainRows;
            // 4 workgroups of 4 waves per CU (128 VGPRs each) = 1024 resident: pick the number of parts per tile so
            // that the grid is a whole number of 1024-workgroup rounds where possible
            uint32_t parts = 1;
#ifdef BT_DEBUG_HOOKS
            if (const char* e = getenv("BT_FUSED_PARTS")) parts = uint32_t(atoi(e));
            else
#endif
            {
                while (parts < chunks && uint64_t(items.size()) * parts < 1024) parts++;
                for (uint32_t cand = parts; cand <= std::min(chunks, parts + 8); cand++)
                    if ((uint64_t(items.size()) * cand) % 1024 == 0) { parts = cand; break; }
            }
            parts = std::max(parts, (chunks + kMaxChunks - 1) / kMaxChunks);  // row tables of a workgroup hold kMaxChunks chunks
            main_job.args.groups = std::max(1u, std::min(parts, chunks));
        }
        {   // LDS window of a workgroup: T consecutive mosaic columns x (kMainRows + 2b) mosaic rows of the source
            double ratio_x = 0.0, ratio_y = 0.0;
            uint64_t max_pitch = 0;
            const double mosaic = double(1u << lod_hi) * double(m.center_size);
            for (const Task* t : splits) {
                const RasterDev& r = p->rasters[t->raster].dev;
                max_pitch = std::max<uint64_t>(max_pitch, r.pitch);
                ratio_x = std::max(ratio_x, double(r.width) / (double(args.brx - args.tlx) * mosaic));
                ratio_y = std::max(ratio_y, double(r.height) / (double(args.bry - args.tly) * mosaic));
            }
            const uint32_t rows = std::min(kMainRows, m.center_size) + 2 * m.border_size;
            const uint64_t cols_needed = uint64_t(double(m.texture_size - 1) * ratio_x) + 4 + 7;
            uint64_t rows_needed = uint64_t(double(rows - 1) * ratio_y) + 4;  // contiguous source-row range (safe bound)
            uint64_t exact_core = rows_needed;  // ... of the centre rows alone (without the first / last chunk's apron rows)
            {   // exact: replay the kernel's own window computation for every tile row and chunk (same f32 operations)
                const uint32_t c = m.center_size, b = m.border_size, chunks = (c + kMainRows - 1) / kMainRows;
                const float scale = float(1u << lod_hi);
                uint64_t exact = 1;
                exact_core = 1;
                std::vector<std::pair<uint32_t, int>> seen;  // (tile row, raster)
                for (const Task* t : splits) {
                    const std::pair<uint32_t, int> key(t->coord.y, t->raster);
                    if (std::find(seen.begin(), seen.end(), key) != seen.end()) continue;
                    seen.push_back(key);
                    const uint32_t H = p->rasters[t->raster].dev.height, n = 1u << lod_hi, ty = t->coord.y;
                    auto axis = [&](uint32_t tile, uint32_t r) { return split_axis(r, c, tile, scale, args.tly, args.bry, H); };
                    for (uint32_t k = 0; k < chunks; k++) {
                        const uint32_t r0 = k * kMainRows, r1 = std::min(c, r0 + kMainRows) - 1;
                        int lo = axis(ty, r0).i0, hi = axis(ty, r1).i1;
                        exact_core = std::max<uint64_t>(exact_core, uint64_t(hi - lo + 1));
                        if (k == 0) lo = std::min(lo, ty > 0 ? axis(ty - 1, c - b).i0 : axis(ty, 0).i0);
                        if (k == chunks - 1) hi = std::max(hi, ty + 1 < n ? axis(ty + 1, b - 1).i1 : axis(ty, c - 1).i1);
                        exact = std::max<uint64_t>(exact, uint64_t(hi - lo + 1));
                    }
                }
                rows_needed = std::min(rows_needed, exact);
            }
            const uint64_t pitch = (cols_needed + 7) / 8 * 8;
            const uint64_t budget = 65536 - sizeof(MainShared);
            main_job.args.lds_pitch = uint32_t(std::min<uint64_t>(pitch, 1u << 20));
            // the whole window must fit (the bounds above are conservative); otherwise lds_rows = 0 selects the
            // kernel variant that reads the source directly
            // the staging loop holds one batch of 8 x 16-byte loads per thread: the window must fit that too
            // and its byte offsets from the first row are kept in 32 bits
            // (the register staging holds one batch of 4 x 16-byte loads per thread: a window of more pieces than that — ratios from ~1.2 up at
            // T = 512 — is staged by LDS-DMA alone, when every raster of the job is 16-byte aligned; rounds 2 - 6 sent such jobs to the unstaged kernel)
            // Four workgroups per CU need <= 40 KB each (160 KB of LDS).  When the window with the apron rows is past that and the centre rows alone are
            // not (ratios ~1.25 - 1.35 at T = 512; at 1.41 the same step takes the job from two workgroups to three), the apron rows — 2b of a tile's T —
            // read global memory and the window shrinks: ratio 1.3, 553 -> 3xx us (round 6, profiles/r06_gebco_size.txt)
            {
                const uint64_t kQuarter = (160u << 10) / 4, kThird = (160u << 10) / 3, fixed = sizeof(MainShared);
                const uint64_t with_aprons = fixed + 2 * rows_needed * pitch * 2, core = fixed + 2 * std::min(rows_needed, exact_core) * pitch * 2;
                auto groups = [&](uint64_t bytes) { return bytes <= kQuarter ? 4 : bytes <= kThird ? 3 : bytes <= (80u << 10) ? 2 : 1; };
                bool all_aligned = true;
                for (const Task* t : splits) {
                    const RasterDev& r = p->rasters[t->raster].dev;
                    if (((reinterpret_cast<uintptr_t>(r.data) | r.pitch) & 15u) != 0) all_aligned = false;
                }
                // (only the run-time-pitch DMA variant knows the mode: aligned rasters, T = 512 or a window past the register batch, not the 528 pitch)
                const uint64_t core_rows = std::min(rows_needed, exact_core);
                const bool dma_variant = all_aligned && pitch <= 4096 && pitch != 528 && (m.texture_size == 512 || core_rows * (pitch / 8) > 256 * 4);
                if (dma_variant && groups(core) > groups(with_aprons)) {
                    main_job.args.apron_global = 1;
                    rows_needed = core_rows;
                }
            }
            uint64_t buffers = 2;
            {   // ... and where even the centre rows alone, twice, leave room for three workgroups or fewer, ONE buffer of them may leave room for four (FusedArgs::single_buffer)
                const uint64_t kQuarter = (160u << 10) / 4, fixed = sizeof(MainShared);
                bool all_aligned = true;
                for (const Task* t : splits) {
                    const RasterDev& r = p->rasters[t->raster].dev;
                    if (((reinterpret_cast<uintptr_t>(r.data) | r.pitch) & 15u) != 0) all_aligned = false;
                }
                const uint64_t core_rows = std::min(rows_needed, exact_core);
                const bool dma_variant = all_aligned && pitch <= 4096 && pitch != 528 && (m.texture_size == 512 || core_rows * (pitch / 8) > 256 * 4);
                if (dma_variant && fixed + 2 * rows_needed * pitch * 2 > kQuarter && fixed + core_rows * pitch * 2 <= kQuarter) {
                    main_job.args.single_buffer = 1;
                    main_job.args.apron_global = 1;
                    rows_needed = core_rows;
                    buffers = 1;
                }
            }
            const bool fits_lds = buffers * rows_needed * pitch * 2 <= budget && (rows_needed + 1) * max_pitch < (1ull << 31);
            const bool fits_batch = rows_needed * (pitch / 8) <= 256 * 4;
            bool aligned = true;
            for (const Task* t : splits) {
                const RasterDev& r = p->rasters[t->raster].dev;
                if (((reinterpret_cast<uintptr_t>(r.data) | r.pitch) & 15u) != 0) aligned = false;
            }
            // (and by choice at T = 512, where it measured 2 % faster than the register staging on a 86400 x 43200 job; at T = 256 — half-empty
            // 1 KB pieces — the register staging is 6 % faster and stays)
            main_job.dma_only = fits_lds && aligned && pitch <= 4096 && (!fits_batch || m.texture_size == 512);
            main_job.args.lds_rows = fits_lds && (fits_batch || main_job.dma_only) ? uint32_t(rows_needed) : 0u;
            if ((main_job.args.apron_global || main_job.args.single_buffer) && main_job.args.lds_rows && !main_job.dma_only) {  // (cannot happen by the conditions above; a window without apron rows in a variant that stages them would overrun)
                main_job.args.apron_global = 0;
                main_job.args.single_buffer = 0;
                main_job.args.lds_rows = 0;
            }
            main_job.dma = main_job.args.lds_rows != 0;
            main_job.args.rotate_priority = 1;
#ifdef BT_DEBUG_HOOKS
            if (const char* e = getenv("BT_FUSED_ROTATE")) main_job.args.rotate_priority = atoi(e) != 0;
#endif
            for (const Task* t : splits) {
                const RasterDev& r = p->rasters[t->raster].dev;
                if (((reinterpret_cast<uintptr_t>(r.data) | r.pitch) & 15u) != 0) main_job.dma = false;
            }
#ifdef BT_DEBUG_HOOKS
            if (const char* e = getenv("BT_FUSED_DMA")) main_job.dma = main_job.dma && (atoi(e) != 0 || main_job.dma_only);
#endif
        }
        Launch lm{};
        lm.kind = kLaunchFusedMain;
        lm.kernels = main_job.args.lds_rows ? 1u : 2u;  // (without the LDS window: fused_corner + fused_main in one entry)
        lm.attachment = ai;
        lm.task_count = uint32_t(items.size());
        lm.aux0 = uint32_t(jobs.size());
        lm.algorithmic_bytes = source_bytes;
        for (uint32_t k = 0; k < main_levels; k++) lm.algorithmic_bytes += tiles_at(lod_hi - k) * Tt * Tt * bpp;
        jobs.push_back(main_job);
        plan.push_back(lm);
        }

        const bool tail_follows = lod_hi - (main_levels - 1) > lod_lo;
        // (after fused_direct, which writes centres only, the tail's extra workgroups do all four sides: no stitch launch)
        const bool rows_in_tail = !shard && tail_follows && main_levels > 1 && (direct || m.border_size % 2u == 0);
        if (main_levels > 1 && !rows_in_tail) {
            // fused_main writes the centres and the left / right apron columns of the parent / grand-parent tiles; their
            // top / bottom apron rows (whole 1 KB rows) come from the batched stitch kernel — sharded: everything, after
            // the all-gather, from the then complete centres
            const uint32_t first = uint32_t(tasks.size());
            for (const Task* t : stitches) {
                if (t->coord.lod == lod_hi || t->coord.lod + main_levels <= lod_hi) continue;
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
            ls.phase = shard ? 2u : 0u;
            // fused_main also writes the left / right apron columns of these tiles (from registers); sharded runs lose
            // the ones that crossed a strip boundary in the all-gather and re-stitch everything
            ls.aux0 = (shard || direct) ? 0u : 1u;  // (fused_direct writes centres only: all four sides)
            if (!shard && !direct) ls.algorithmic_bytes = uint64_t(ls.task_count) * 2 * (2 * m.border_size * Tt) * bpp;
            if (ls.task_count) plan.push_back(ls);
        }

        // tail launches: three LODs at a time below the last fused one
        uint32_t in_lod = lod_hi - (main_levels - 1);
        int first_tail_job = -1, first_tail_plan = -1;
        uint32_t tail_launches = 0;
        while (in_lod > lod_lo) {
            tail_launches++;
            const uint32_t levels = std::min(3u, in_lod - lod_lo);
            uint64_t lt_extra = 0;
            FusedJobDev tail{args, ai};
            tail.all_layers = all_layers;
            tail.args.lod = in_lod;
            tail.args.levels = levels;
            // the first tail launch also fills the top / bottom apron rows of the LODs fused_main produced (see above)
            tail.args.apron_lods = (rows_in_tail && in_lod == lod_hi - (main_levels - 1)) ? main_levels - 1 : 0u;
            tail.args.apron_cols = direct ? 1u : 0u;
            if (tail.args.apron_lods)
                for (uint32_t k = 0; k < tail.args.apron_lods; k++) lt_extra += tiles_at(in_lod + k) * 2 * (2 * m.border_size * (Tt + (direct ? cc : 0))) * bpp;
            Launch lt{};
            lt.kind = kLaunchFusedTail;
            lt.attachment = ai;
            lt.aux0 = uint32_t(jobs.size());
            lt.phase = shard ? 2u : 0u;
            lt.algorithmic_bytes = tiles_at(in_lod) * cc * cc * bpp + lt_extra;
            for (uint32_t k = 1; k <= levels; k++) {
                lt.algorithmic_bytes += tiles_at(in_lod - k) * Tt * Tt * bpp;
                lt.task_count += uint32_t(tiles_at(in_lod - k));
            }
            if (first_tail_job < 0) {
                first_tail_job = int(jobs.size());
                first_tail_plan = int(plan.size());
            }
            jobs.push_back(tail);
            plan.push_back(lt);
            in_lod -= levels;
        }

        // cube: aprons that cross a face edge (stitch.wgsl:12-51, 79-118), one task — one workgroup — per region: only the apron regions whose
        // neighbour lives on another face (the fused kernels wrote the rest).  The regions of the LODs fused_main produced read centres that are
        // complete when the tail launch starts: they ride in that launch as extra workgroups (round 5: the 16k-texel-wide job's seam launch was
        // 33 of 500 us), and the tail's own apron-row workgroups leave those regions alone (FusedArgs::seam_skip).  What the tail itself
        // produces (the few tiles of the top LODs) is stitched by the generic kernel behind it, as before.
        if (spherical) {
            auto seam_task = [&](const Task* t) {
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
                return d;
            };
            auto on_face_edge = [](const Task* t) {
                const uint32_t n = 1u << t->coord.lod;
                return t->coord.x == 0 || t->coord.y == 0 || t->coord.x == n - 1 || t->coord.y == n - 1;
            };
            // in the tail launch: R16 main plan, unsharded, a tail launch with apron-row workgroups exists, and EVERY region beyond exactly one
            // face edge of the tiles whose apron rows the tail writes has its neighbour (then "skip" and "a seam workgroup writes it" coincide)
            bool in_tail = !shard && !hybrid && first_tail_job >= 0 && jobs[size_t(first_tail_job)].args.apron_lods != 0;  // (round 6: Rgba8 after fused_direct too)
#ifdef BT_DEBUG_HOOKS
            if (getenv("BT_FUSED_SEAMS_LATE")) in_tail = false;
#endif
            const uint32_t main_lo = lod_hi - (main_levels - 1);  // LODs main_lo .. lod_hi come out of the main launch
            if (in_tail)
                for (const Task* t : stitches) {
                    if (!on_face_edge(t) || t->coord.lod < main_lo || t->coord.lod >= lod_hi) continue;
                    const int n = int(1u << t->coord.lod);
                    static const int off[8][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};  // coordinate.rs:209-218
                    for (int i = 0; i < 8; i++) {
                        const int nx = int(t->coord.x) + off[i][0], ny = int(t->coord.y) + off[i][1];
                        const bool out_x = nx < 0 || nx >= n, out_y = ny < 0 || ny >= n;
                        if (out_x != out_y && !(t->rel[i].atlas_index != BT_INVALID_ATLAS_INDEX && t->rel[i].coordinate.side != t->coord.side)) in_tail = false;
                    }
                }
            if (in_tail) {
                // ... and every face-edge tile of those LODs that the grids hold HAS a stitch task (seam_skip makes the tail's apron-row workgroups
                // leave the cross-face regions of every such grid tile alone: one without a task would keep stale apron texels there)
                std::unordered_set<uint32_t> stitched;
                for (const Task* t : stitches)
                    if (on_face_edge(t) && t->coord.lod >= main_lo && t->coord.lod < lod_hi) stitched.insert(t->atlas_index);
                for (uint32_t side = 0; side < sides && in_tail; side++)
                    for (uint32_t lod = main_lo; lod < lod_hi && in_tail; lod++) {
                        const uint32_t n = 1u << lod, off = grid_offsets[side * 32 + lod];
                        for (uint32_t x = 0; x < n && in_tail; x++)
                            for (uint32_t y = 0; y < n; y++) {
                                if (!(x == 0 || y == 0 || x == n - 1 || y == n - 1)) continue;
                                const uint32_t v = grids[off + (size_t(x) << lod) + y];
                                if (v != kInvalid && !stitched.count(v)) { in_tail = false; break; }
                            }
                    }
            }
            // Round 6: the cross-face regions of the LODs the tail ITSELF produces ride in it as well, PULLED from the tail's input (FusedArgs::seam_pull) —
            // when ONE tail launch produces every lower LOD (its input then lies at most three LODs above any of them), every region beyond exactly
            // one face edge of those tiles has its cross-face neighbour, and every face-edge grid tile of those LODs has a stitch task (the tail's
            // pushes leave those regions alone).  No stitch launch follows the tail then: the cube job is two launches.
            bool pull = in_tail && tail_launches == 1;
#ifdef BT_DEBUG_HOOKS
            if (getenv("BT_FUSED_NO_PULL")) pull = false;
#endif
            if (pull) {
                std::unordered_set<uint32_t> stitched;
                for (const Task* t : stitches) {
                    if (!on_face_edge(t) || t->coord.lod >= main_lo) continue;
                    stitched.insert(t->atlas_index);
                    const int n = int(1u << t->coord.lod);
                    static const int off[8][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
                    for (int i = 0; i < 8; i++) {
                        const int nx = int(t->coord.x) + off[i][0], ny = int(t->coord.y) + off[i][1];
                        const bool out_x = nx < 0 || nx >= n, out_y = ny < 0 || ny >= n;
                        const bt_atlas_tile& r = t->rel[i];
                        if (out_x != out_y && !(r.atlas_index != BT_INVALID_ATLAS_INDEX && r.coordinate.side != t->coord.side && r.coordinate.lod == t->coord.lod &&
                                                r.coordinate.x < 65536u && r.coordinate.y < 65536u))
                            pull = false;
                    }
                }
                for (uint32_t side = 0; side < sides && pull; side++)
                    for (uint32_t lod = lod_lo; lod < main_lo && pull; lod++) {
                        const uint32_t n = 1u << lod, off = grid_offsets[side * 32 + lod];
                        for (uint32_t x = 0; x < n && pull; x++)
                            for (uint32_t y = 0; y < n; y++) {
                                if (!(x == 0 || y == 0 || x == n - 1 || y == n - 1)) continue;
                                const uint32_t v = grids[off + (size_t(x) << lod) + y];
                                if (v != kInvalid && !stitched.count(v)) { pull = false; break; }
                            }
                    }
            }
            uint64_t tail_pixels = 0, late_pixels = 0;
            const uint32_t tail_first = uint32_t(tasks.size());
            for (int pass = 0; pass < 2; pass++) {  // the tail launch's regions first, then the later launch's
                if (pass == 1 && in_tail) {
                    FusedJobDev& tj = jobs[size_t(first_tail_job)];
                    tj.seam_first = tail_first;
                    tj.args.seam_count = uint32_t(tasks.size()) - tail_first;
                    tj.args.seam_skip = 1;
                    tj.args.seam_pull = pull ? 1u : 0u;
                    plan[size_t(first_tail_plan)].algorithmic_bytes += 2 * tail_pixels * bpp;
                }
                const uint32_t first = uint32_t(tasks.size());
                // (pass 0 in two sweeps: the pulled regions FIRST — they are the launch's longest workgroups and the seam list's head is dispatched first)
                for (int sweep = (pass == 0 && pull) ? 0 : 1; sweep < 2; sweep++)
                for (const Task* t : stitches) {
                    if (!on_face_edge(t)) continue;
                    if (hybrid && t->coord.lod == lod_hi) continue;  // stitched completely by the batched kernel above
                    const bool pulled = pull && t->coord.lod < main_lo;
                    const bool rides = in_tail && (t->coord.lod >= main_lo || pulled);
                    if (rides != (pass == 0)) continue;
                    if (pass == 0 && pull && pulled != (sweep == 0)) continue;
                    const TaskDev d = seam_task(t);
                    for (int i = 0; i < 8; i++)
                        if (d.rel_index[i] != BT_INVALID_ATLAS_INDEX && d.rel_side[i] != d.side) {
                            TaskDev e = d;
                            e.regions = 1u << i;
                            uint32_t parts = 1;
                            if (pulled) {  // evaluated from the tail's input LOD (main_lo) on the neighbour face: how many LODs up, and where the neighbour tile lies
                                // an edge region is shared out so that a thread evaluates one pixel pair (256 per workgroup); a corner region is one workgroup
                                const uint32_t pixels = m.border_size * (i < 4 ? m.center_size : m.border_size);
                                const uint32_t pairs = (m.format != BT_FORMAT_R16 || ((m.border_size | m.texture_size) & 1u)) ? pixels : pixels / 2u;  // what one thread stores
                                parts = std::max(1u, std::min(255u, (pairs + 255u) / 256u));
                                e.rel_index[i] = (t->rel[i].coordinate.x << 16) | t->rel[i].coordinate.y;
                            }
                            for (uint32_t part = 0; part < parts; part++) {
                                if (pulled) e.raster = (main_lo - t->coord.lod) | (part << 8) | (parts << 16);
                                tasks.push_back(e);
                            }
                            (pass == 0 ? tail_pixels : late_pixels) += uint64_t(m.border_size) * (i < 4 ? cc : m.border_size);
                        }
                }
                if (pass == 0) continue;
                Launch ls{};
                ls.kind = kLaunchStitch;
                ls.aux0 = 2u;  // one region per task
                ls.attachment = ai;
                ls.first_task = first;
                ls.task_count = uint32_t(tasks.size()) - first;
                ls.algorithmic_bytes = 2 * late_pixels * bpp;
                ls.phase = shard ? 2u : 0u;
                if (ls.task_count) plan.push_back(ls);
            }
        }
    }
    return true;
}

bt_status fused_launch_range(bt_preprocessor* p, bt_atlas* a, const Launch& l, uint32_t item_begin, uint32_t item_count);
bt_status fused_launch(bt_preprocessor* p, bt_atlas* a, const Launch& l) { return fused_launch_range(p, a, l, 0u, 0xFFFFFFFFu); }

// Before the launches of a run, in plan order: a fused main / direct launch whose finest tiles are all still unwritten since bt_atlas_create
// (Attachment::written) runs with FusedArgs::prev_zero — bit-identical by construction, the fetch would return the memset's 0 — and every
// launch marks the layers it writes, so a later job of the same queue that overlays these tiles takes the fetching path.  Returns how many
// launches got the flag (bt_run_stats::prev_zero_launches).
uint32_t fused_begin_run(bt_preprocessor* p, bt_atlas* a) {
    uint32_t flagged = 0;
    for (const Launch& l : p->plan) {
        Attachment& at = a->attachments[l.attachment];
        if (l.kind == kLaunchSplit || l.kind == kLaunchDownsample || l.kind == kLaunchStitch) {
            for (uint32_t i = l.first_task; i < l.first_task + l.task_count && i < p->tasks_host.size(); i++) at.mark_written(p->tasks_host[i].atlas_index, 1);
            continue;
        }
        if (!p->fused || l.aux0 >= p->fused->jobs.size()) continue;
        FusedJobDev& job = p->fused->jobs[l.aux0];
        if (l.kind == kLaunchFusedMain || l.kind == kLaunchFusedDirect) {
            bool fresh = !job.finest_layers.empty();
            for (uint32_t layer : job.finest_layers) fresh = fresh && layer < at.written.size() && !at.written[layer];
#ifdef BT_DEBUG_HOOKS
            if (getenv("BT_FUSED_NO_PREV_ZERO")) fresh = false;
#endif
            job.args.prev_zero = fresh ? 1u : 0u;
            flagged += fresh;
        }
        for (uint32_t layer : job.all_layers) at.mark_written(layer, 1);
    }
    return flagged;
}

// Bands of whole tile rows of a fused main launch, with the last source row each band's kernels read (the bottom apron rows
// of its last tile row are evaluated with the next tile row's formula: same f32 operations as the kernel's row tables).
// The source texels [x0, x1) x [y0, y1) of raster `raster` that the launches of the compiled plan read — for a sharded
// preprocessor: this rank's column strips + their halo (finest aprons are evaluated from the source; fused_main's staged windows
// start on an 8-texel boundary and are lds_pitch wide; fused_direct reads texel by texel).  Conservative (a missing neighbour tile
// only shrinks what the kernel reads).  Decided PER RASTER: false — the caller assumes the whole raster — unless every launch that
// reads it is a fused_main / fused_direct launch with an item list (a height raster read by fused_main and an albedo raster read by
// fused_direct in one queue get a window each; a raster the hybrid plan's batched split reads has none).
bool fused_source_window(const bt_preprocessor* p, uint32_t raster, uint32_t out[4]) {
    if (!p->fused || raster >= p->rasters.size()) return false;
    if (raster < p->fused->whole_raster.size() && p->fused->whole_raster[raster]) return false;
    const RasterDev& r = p->rasters[raster].dev;
    uint32_t x0 = r.width, y0 = r.height, x1 = 0, y1 = 0;
    bool any = false, known = false;
    for (const FusedJobDev& job : p->fused->jobs) {
        if (job.host_items.empty()) continue;
        known = true;
        const FusedArgs& A = job.args;
        const uint32_t c = A.m.center_size, b = A.m.border_size, n = 1u << A.lod;
        const float scale = float(n);
        auto ax = [&](uint32_t tile, uint32_t col) { return split_axis(col, c, tile, scale, A.tlx, A.brx, r.width); };
        auto ay = [&](uint32_t tile, uint32_t row) { return split_axis(row, c, tile, scale, job.tly, job.bry, r.height); };
        for (const MainItem& it : job.host_items) {
            if (it.raster != raster) continue;
            any = true;
            const int lo_x = std::min(it.x > 0 ? ax(it.x - 1, c - b).i0 : ax(it.x, 0).i0, ax(it.x, 0).i0);
            const int hi_x = std::max(it.x + 1 < n ? ax(it.x + 1, b - 1).i1 : ax(it.x, c - 1).i1, ax(it.x, c - 1).i1);
            const int lo_y = std::min(it.y > 0 ? ay(it.y - 1, c - b).i0 : ay(it.y, 0).i0, ay(it.y, 0).i0);
            const int hi_y = std::max(it.y + 1 < n ? ay(it.y + 1, b - 1).i1 : ay(it.y, c - 1).i1, ay(it.y, c - 1).i1);
            uint32_t xa = uint32_t(std::max(lo_x, 0)), xe = uint32_t(hi_x) + 1u;
            if (!job.direct) {
                xa &= ~7u;
                if (A.lds_rows) xe = std::max(xe, xa + A.lds_pitch);  // the staged window: lds_pitch texels from the aligned start
                if (xe + 8u > r.width) xe = r.width;                   // (pieces past the row's end re-read its last 16 bytes)
            }
            x0 = std::min(x0, xa);
            x1 = std::max(x1, std::min(xe, r.width));
            y0 = std::min(y0, uint32_t(std::max(lo_y, 0)));
            y1 = std::max(y1, std::min(uint32_t(hi_y) + 1u, r.height));
        }
    }
    if (!known) return false;
    if (!any) x0 = y0 = x1 = y1 = 0;  // a cube face none of this rank's units lie on
    out[0] = x0;
    out[1] = y0;
    out[2] = x1;
    out[3] = y1;
    return true;
}

// Bands of whole tile rows of a fused main / direct launch (streamed runs), with the raster each band reads and the last source row its
// kernels touch (the bottom apron rows of its last tile row are evaluated with the next tile row's formula: same f32 operations as the
// kernels' row tables).  The items are in (side, tile row, x) order; a band never crosses a side (a cube job: six rasters, one after the
// other).  tile_rows_per_band == 0: a quarter of the face's tile rows, at most 4 (the 16k job: 8 bands of 4 rows; a 4k job: 4 bands of 2).
bool fused_stream_bands(bt_preprocessor* p, const Launch& l, uint32_t tile_rows_per_band, std::vector<StreamBand>* bands) {
    if ((l.kind != kLaunchFusedMain && l.kind != kLaunchFusedDirect) || !p->fused || l.aux0 >= p->fused->jobs.size()) return false;
    const FusedJobDev& job = p->fused->jobs[l.aux0];
    const std::vector<MainItem>& items = job.host_items;
    if (items.empty() || (l.kind == kLaunchFusedMain && job.args.lds_rows == 0)) return false;  // (the unstaged fused_main is two kernels over the whole item list)
    const uint32_t c = job.args.m.center_size, b = job.args.m.border_size, n = 1u << job.args.lod;
    if (tile_rows_per_band == 0) tile_rows_per_band = std::max(1u, std::min(4u, n / 4u));
    const float scale = float(n);
    bands->clear();
    size_t i = 0;
    while (i < items.size()) {
        const uint32_t side = items[i].side, raster = items[i].raster;
        if (raster >= p->rasters.size()) return false;
        const RasterDev& r = p->rasters[raster].dev;
        auto axis = [&](uint32_t tile, uint32_t row) { return split_axis(row, c, tile, scale, job.tly, job.bry, r.height); };
        StreamBand band{};
        band.item_begin = uint32_t(i);
        band.tile_y_begin = items[i].y;
        band.side = side;
        band.raster = raster;
        uint32_t rows = 0, y = items[i].y;
        while (i < items.size() && items[i].side == side && (items[i].y == y || rows + 1 < tile_rows_per_band)) {
            if (items[i].raster != raster) return false;  // one source per side
            if (items[i].y != y) {
                if (items[i].y < y) return false;  // not in tile-row order
                y = items[i].y;
                rows++;
            }
            i++;
        }
        band.item_count = uint32_t(i) - band.item_begin;
        band.tile_y_end = y + 1;
        const int hi = y + 1 < n ? axis(y + 1, b - 1).i1 : axis(y, c - 1).i1;
        band.source_row_end = uint32_t(std::min<int64_t>(int64_t(r.height), int64_t(hi) + 1));
        // a raster's bands follow each other and its source rows never decrease from band to band (they do not for a dataset rectangle with
        // top < bottom); a raster that comes back after another one's bands cannot be streamed
        if (!bands->empty() && bands->back().raster == raster) {
            if (bands->back().source_row_end > band.source_row_end) return false;
        } else {
            for (const StreamBand& earlier : *bands)
                if (earlier.raster == raster) return false;
        }
        bands->push_back(band);
    }
    return true;
}

void fused_launch_tiles(const bt_preprocessor* p, const Launch& l, uint32_t item_begin, uint32_t item_count, std::vector<FusedTile>* out) {
    out->clear();
    if ((l.kind != kLaunchFusedMain && l.kind != kLaunchFusedDirect) || !p->fused || l.aux0 >= p->fused->jobs.size()) return;
    const FusedJobDev& job = p->fused->jobs[l.aux0];
    for (uint64_t i = item_begin; i < uint64_t(item_begin) + item_count && i < job.host_items.size(); i++) {
        const MainItem& it = job.host_items[size_t(i)];
        out->push_back({{it.side, job.args.lod, it.x, it.y}, it.atlas_index});
    }
}

bt_status fused_launch_range(bt_preprocessor* p, bt_atlas* a, const Launch& l, uint32_t item_begin, uint32_t item_count) {
    (void)a;
    if (!p->fused || l.aux0 >= p->fused->jobs.size()) {
        set_error("fused launch without a plan");
        return BT_ERR_INVALID_ARGUMENT;
    }
    std::vector<FusedJobDev>& jobs = p->fused->jobs;
    FusedJobDev job = jobs[l.aux0];
    job.args.rasters = p->rasters_dev;
    if ((l.kind == kLaunchFusedMain || l.kind == kLaunchFusedDirect) && item_begin < job.args.item_count) {  // a band of the item list (streamed runs); default: all
        job.args.items += item_begin;
        job.args.item_count = std::min(item_count, job.args.item_count - item_begin);
    }
    if (l.kind == kLaunchFusedDirect) {
        const uint32_t blocks_per_tile = (job.args.m.center_size + kDirectRows - 1) / kDirectRows;
        const uint32_t wgs_per_tile = (blocks_per_tile + job.args.groups - 1) / job.args.groups;
        if (job.direct_rep)  // a source coarser than the tile grid: rows repeat the pair above, the chained path follows (round 6)
            fused_direct_rgba8_kernel<true><<<job.args.item_count * wgs_per_tile, 256, 0, p->ctx->stream>>>(job.args);
        else if (job.direct_skips && !job.direct_rep)  // a source finer than the tile grid: rows pass over source rows, the chained path follows with two extra (plain) loads per block
            fused_direct_rgba8_kernel<false, true><<<job.args.item_count * wgs_per_tile, 256, 0, p->ctx->stream>>>(job.args);
        else
            fused_direct_rgba8_kernel<false><<<job.args.item_count * wgs_per_tile, 256, 0, p->ctx->stream>>>(job.args);
    } else if (l.kind == kLaunchFusedMain) {
        const uint32_t blocks = job.args.item_count * job.args.groups;
        if (job.args.lds_rows) {
            size_t lds = sizeof(MainShared) + (job.args.single_buffer ? 1 : 2) * size_t(job.args.lds_rows) * job.args.lds_pitch * 2;
            lds = std::min<size_t>(65536, lds + job.lds_pad);  // (occupancy experiments)
            if (job.args.m.texture_size == 512 && job.args.lds_pitch == 528 && job.dma)
                fused_main_kernel<true, false, 512, 528, true><<<blocks, 256, lds, p->ctx->stream>>>(job.args);
            else if (job.dma_only)  // a window the register staging cannot batch (a source-to-tile ratio away from 1), or T = 512 at any other pitch: LDS-DMA with a run-time pitch (round 6)
                fused_main_kernel<true, false, 0, 0, true><<<blocks, 256, lds, p->ctx->stream>>>(job.args);
            else if (job.args.m.texture_size == 512 && job.args.lds_pitch == 528)
                fused_main_kernel<true, false, 512, 528><<<blocks, 256, lds, p->ctx->stream>>>(job.args);
            else
                fused_main_kernel<true, false, 0, 0><<<blocks, 256, lds, p->ctx->stream>>>(job.args);
        } else {
            fused_corner_kernel<<<job.args.item_count, 64, 0, p->ctx->stream>>>(job.args);
            fused_main_kernel<false, true, 0, 0><<<blocks, 256, sizeof(MainShared), p->ctx->stream>>>(job.args);
        }
    } else {
        const uint32_t size = (1u << job.args.lod) * job.args.m.center_size, nx = (size + 63) / 64;
        job.args.seam_tasks = p->tasks_dev + job.seam_first;  // ((re)allocated with the plan: patched at launch like the rasters)
        uint64_t extras = 0;  // apron blocks per side
        if (job.args.apron_lods) {
            const uint32_t blocks_per_tile = job.args.m.format == BT_FORMAT_R16
                ? (job.args.m.border_size * job.args.m.texture_size + 256u * kApronPairsPerThread - 1u) / (256u * kApronPairsPerThread)  // texel pairs of the 2b apron rows
                : (2u * job.args.m.border_size * (job.args.m.texture_size + (job.args.apron_cols ? job.args.m.center_size : 0u)) + 255u) / 256u;
            for (uint32_t k = 0; k < job.args.apron_lods; k++) extras += (1ull << (2 * (job.args.lod + k))) * blocks_per_tile;
        }
        job.args.tail_extras = uint32_t(extras);
        // a 1-D grid: XCD k (blockIdx.x % 8) takes its share of the seam regions, then the k-th eighth of the apron blocks, then the k-th eighth of the mosaic (see the kernel)
        const uint64_t total_m = uint64_t(job.args.sides) * nx * nx, total_a = uint64_t(job.args.sides) * extras;
        const uint32_t blocks = uint32_t(8 * ((total_m + 7) / 8 + (total_a + 7) / 8 + (uint64_t(job.args.seam_count) + 7) / 8));
        if (job.args.m.format == BT_FORMAT_R16) {
            if (job.args.regular) fused_tail_kernel<BT_FORMAT_R16, true><<<blocks, 256, 0, p->ctx->stream>>>(job.args);
            else fused_tail_kernel<BT_FORMAT_R16, false><<<blocks, 256, 0, p->ctx->stream>>>(job.args);
        } else {
            if (job.args.regular) fused_tail_kernel<BT_FORMAT_RGBA8, true><<<blocks, 256, 0, p->ctx->stream>>>(job.args);
            else fused_tail_kernel<BT_FORMAT_RGBA8, false><<<blocks, 256, 0, p->ctx->stream>>>(job.args);
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "fused kernel launch");
    return BT_OK;
}

}  // namespace bt

extern "C" bt_status bt_selftest(bt_ctx* ctx, uint32_t* failures) {
    if (!ctx || !failures) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    uint32_t* dev = nullptr;
    BT_HIP(hipMalloc((void**)&dev, sizeof(uint32_t)));
    hipError_t e = hipMemsetAsync(dev, 0, sizeof(uint32_t), ctx->stream);
    if (e == hipSuccess) {
        bt::selftest_kernel<<<256, 256, 0, ctx->stream>>>(dev);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(failures, dev, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    hipFree(dev);
    if (e != hipSuccess) return bt::hip_fail(e, "bt_selftest");
    return BT_OK;
}
