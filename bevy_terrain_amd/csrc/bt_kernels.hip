// Reference-shaped batched kernels: one launch per queue phase instead of one dispatch per tile.
//   split      <- shaders/preprocess/split.wgsl:18-43
//   downsample <- shaders/preprocess/downsample.wgsl:12-40
//   stitch     <- shaders/preprocess/stitch.wgsl:12-118
//   mip level  <- terrain_data/mod.rs:143-219
// These are the general path (any dataset rectangle, any tile-existence pattern, cube faces, both
// formats); the fused split+pyramid kernels in bt_fused.hip cover the common case faster.
//
// Arithmetic contract (identical to oracle/bt_oracle.c, checked bit-for-bit by tests/): IEEE binary32,
// one rounding per written operation — this file is compiled with -ffp-contract=off; `/` and sqrtf are
// correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt).
//
// Unlike the WGSL these kernels work in place on the atlas layers: the reference's write-section round
// trip (preprocess/mod.rs:169-210) exists only because a storage texture cannot be read and written in
// one pass.  split reads only its own texel's previous value, downsample reads child layers, stitch
// reads centres and writes aprons, so no task of a phase reads what another task of that phase writes.
#include "bt_internal.hpp"

namespace bt {

namespace {

constexpr uint32_t kInvalid = 0xFFFFFFFFu;

// t / 65535 and t / 255, correctly rounded, as mul + two fma (Markstein): equal to the IEEE division for every
// input of the range — exhaustively checked on the device by bt_selftest and on the CPU by the oracle tests
__device__ __forceinline__ float unorm16_to_float(uint32_t t) {
    const float x = float(t), r = 1.0f / 65535.0f;
    const float q0 = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q0, 65535.0f, x), r, q0);
}
__device__ __forceinline__ float unorm8_to_float(uint32_t t) {
    const float x = float(t), r = 1.0f / 255.0f;
    const float q0 = x * r;
    return __builtin_fmaf(__builtin_fmaf(-q0, 255.0f, x), r, q0);
}

// pack2x16unorm / pack4x8unorm component: floor(0.5 + N * clamp(e, 0, 1))
__device__ __forceinline__ uint32_t float_to_unorm(float e, float n) {
    const float cl = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
    return uint32_t(floorf(0.5f + n * cl));
}

// WGSL mix(a, b, t) = a * (1 - t) + b * t
__device__ __forceinline__ float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

struct Axis {
    int i0, i1;  // clamped texel indices of the bilinear footprint
    float fr;    // fractional weight
};

// One axis of split.wgsl:25-32 for texture pixel p of tile index `tile`:
// tile_coords = (p - b) / c ; source = (tile + tile_coords) / 2^lod ; uv = inverse_mix(lo, hi, source);
// sampler: texel centres at uv*dim - 0.5, clamp-to-edge.
__device__ __forceinline__ Axis split_axis(uint32_t p, uint32_t b, uint32_t c, uint32_t tile, float scale, float lo,
                                           float hi, uint32_t dim) {
    const float tc = float(p - b) / float(c);
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

template <uint32_t FORMAT>
struct Texel;

template <>
struct Texel<BT_FORMAT_R16> {
    using type = uint16_t;
    static constexpr uint32_t kPerEntry = 2;
};
template <>
struct Texel<BT_FORMAT_RGBA8> {
    using type = uint32_t;
    static constexpr uint32_t kPerEntry = 1;
};

__device__ __forceinline__ bool is_border(uint32_t px, uint32_t py, uint32_t b, uint32_t c) {
    return !(px >= b && px < b + c && py >= b && py < b + c);
}

// ------------------------------------------------------------------------------------------ split

// The horizontal half of the bilinear filter for one column and one source row: mix(t(i0), t(i1), fx) per channel, and
// whether both texels carry data (textureGather(0, ..) != 0 on channel 0, split.wgsl:34).  A source row is shared by two
// consecutive output rows (its i1 is the next row's i0), so the sweep down a column keeps the last one.
template <uint32_t FORMAT>
struct HRow {
    float h[FORMAT == BT_FORMAT_R16 ? 1 : 4];
    bool valid;
};

// the two raw texels of a column in one source row (global address space: plain global_load, not flat)
template <uint32_t FORMAT>
__device__ __forceinline__ uint2 split_fetch(const RasterDev& r, const Axis& ax, int y) {
    typedef const uint8_t __attribute__((address_space(1))) * global_bytes;
    const global_bytes row = (global_bytes)r.data + uint64_t(y) * r.pitch;
    if constexpr (FORMAT == BT_FORMAT_R16) {
        typedef const uint16_t __attribute__((address_space(1))) * global_u16;
        return make_uint2(((global_u16)row)[ax.i0], ((global_u16)row)[ax.i1]);
    } else {
        typedef const uint32_t __attribute__((address_space(1))) * global_u32;
        return make_uint2(((global_u32)row)[ax.i0], ((global_u32)row)[ax.i1]);
    }
}

template <uint32_t FORMAT>
__device__ __forceinline__ HRow<FORMAT> split_hrow(uint2 t, float fx) {
    HRow<FORMAT> o;
    if constexpr (FORMAT == BT_FORMAT_R16) {
        o.valid = t.x != 0 && t.y != 0;
        o.h[0] = mixf(unorm16_to_float(t.x), unorm16_to_float(t.y), fx);
    } else {
        o.valid = (t.x & 0xFFu) != 0 && (t.y & 0xFFu) != 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) o.h[k] = mixf(unorm8_to_float((t.x >> (8 * k)) & 0xFFu), unorm8_to_float((t.y >> (8 * k)) & 0xFFu), fx);
    }
    return o;
}

// value of one centre pixel as a texel (u16 or packed rgba8) from its two filtered source rows.  Where the footprint
// has no data the pixel keeps what the atlas holds (split.wgsl:37-42): the caller simply does not store it.
template <uint32_t FORMAT>
__device__ __forceinline__ uint32_t split_texel(const HRow<FORMAT>& top, const HRow<FORMAT>& bot, float fy) {
    if constexpr (FORMAT == BT_FORMAT_R16) {
        return float_to_unorm(mixf(top.h[0], bot.h[0], fy), 65535.0f);
    } else {
        uint32_t out = 0;
#pragma unroll
        for (uint32_t k = 0; k < 4; k++) out |= float_to_unorm(mixf(top.h[k], bot.h[k], fy), 255.0f) << (8 * k);
        return out;
    }
}

// grid.x = tasks * ceil(T / kRows); 256 threads sweep kRows rows of one tile, one 32-bit entry per step
// (the reference's thread = one u32 entry, preprocessing.wgsl:59-90)
constexpr uint32_t kRows = 8;
// downsample: every texel costs a chain of dependent loads (task -> child texels), so its workgroups take fewer rows —
// more of them in flight instead of 16 serial round trips per thread
// (only for small launches: with many tiles the 8-row workgroups stream better)

template <uint32_t FORMAT>
__global__ __launch_bounds__(256) void split_kernel(AttachmentMeta m, void* __restrict__ atlas,
                                                    const TaskDev* __restrict__ tasks,
                                                    const RasterDev* __restrict__ rasters, uint32_t row_blocks) {
    using T = typename Texel<FORMAT>::type;
    constexpr uint32_t kPer = Texel<FORMAT>::kPerEntry;
    const uint32_t task_index = blockIdx.x / row_blocks;
    const uint32_t row0 = (blockIdx.x % row_blocks) * kRows;
    const TaskDev task = tasks[task_index];
    const RasterDev raster = rasters[task.raster];
    const uint32_t Tsz = m.texture_size, b = m.border_size, c = m.center_size;
    const uint32_t entries_per_row = Tsz / kPer;
    const float scale = float(1u << task.lod);  // tile_count(lod), functions.wgsl:156
    T* tile = (T*)atlas + uint64_t(task.atlas_index) * Tsz * Tsz;

    const uint32_t rows = min(kRows, Tsz - row0);
    // y parameters once per row (threads 0..rows-1), x parameters once per thread and column: the only divisions
    __shared__ Axis s_ay[kRows];
    __shared__ int s_consecutive;  // all kRows rows are centre rows and step through the source one row at a time
    if (threadIdx.x < rows) {
        const uint32_t py = row0 + threadIdx.x;
        if (py >= b && py < b + c) s_ay[threadIdx.x] = split_axis(py, b, c, task.y, scale, task.tly, task.bry, raster.height);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        bool ok = rows == kRows && row0 >= b && row0 + kRows <= b + c;
        for (uint32_t r = 0; ok && r < kRows; r++) ok = s_ay[r].i0 == s_ay[0].i0 + int(r) && s_ay[r].i1 == s_ay[r].i0 + 1;
        s_consecutive = ok ? 1 : 0;
    }
    __syncthreads();
    const bool consecutive = __builtin_amdgcn_readfirstlane(s_consecutive) != 0;
    auto store = [&](uint32_t py, uint32_t ex, const uint32_t (&texels)[kPer], const bool (&keep)[kPer]) {
        // a pixel whose footprint has no data keeps what the atlas holds (split.wgsl:37-42): it is not stored
        if (py >= m.row_limit) return;  // (BT_RUN_REFERENCE_DISPATCH: rows the reference never dispatches)
        if constexpr (FORMAT == BT_FORMAT_R16) {
            if (!keep[0] && !keep[1]) ((uint32_t*)tile)[(uint64_t(py) * Tsz) / 2 + ex] = texels[0] | (texels[1] << 16);
            else if (!keep[0]) tile[uint64_t(py) * Tsz + 2 * ex] = T(texels[0]);
            else if (!keep[1]) tile[uint64_t(py) * Tsz + 2 * ex + 1] = T(texels[1]);
        } else {
            if (!keep[0]) tile[uint64_t(py) * Tsz + ex] = texels[0];
        }
    };
    for (uint32_t ex = threadIdx.x; ex < entries_per_row; ex += blockDim.x) {
        Axis ax[kPer];
        bool col_is_centre[kPer];
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            const uint32_t px = ex * kPer + k;
            col_is_centre[k] = px >= b && px < b + c;
            ax[k] = col_is_centre[k] ? split_axis(px, b, c, task.x, scale, task.tlx, task.brx, raster.width) : Axis{};
        }
        if (consecutive) {
            // the common case: kRows + 1 consecutive source rows feed the kRows output rows.  All their texels are
            // requested up front (one round of memory latency per column instead of one per row); every source row is
            // filtered horizontally once and used by two output rows.
            const int y_first = __builtin_amdgcn_readfirstlane(s_ay[0].i0);
            uint2 raw[kRows + 1][kPer];
#pragma unroll
            for (uint32_t j = 0; j <= kRows; j++)
#pragma unroll
                for (uint32_t k = 0; k < kPer; k++) raw[j][k] = col_is_centre[k] ? split_fetch<FORMAT>(raster, ax[k], y_first + int(j)) : make_uint2(1u, 1u);
            HRow<FORMAT> top[kPer];
#pragma unroll
            for (uint32_t k = 0; k < kPer; k++) top[k] = split_hrow<FORMAT>(raw[0][k], ax[k].fr);
#pragma unroll
            for (uint32_t r = 0; r < kRows; r++) {
                const float fy = s_ay[r].fr;
                uint32_t texels[kPer];
                bool keep[kPer];
#pragma unroll
                for (uint32_t k = 0; k < kPer; k++) {
                    const HRow<FORMAT> bot = split_hrow<FORMAT>(raw[r + 1][k], ax[k].fr);
                    keep[k] = col_is_centre[k] && !(top[k].valid && bot.valid);
                    texels[k] = col_is_centre[k] ? split_texel<FORMAT>(top[k], bot, fy) : 0u;  // apron columns: zero until stitch
                    top[k] = bot;
                }
                store(row0 + r, ex, texels, keep);
            }
            continue;
        }
        for (uint32_t r = 0; r < rows; r++) {
            const uint32_t py = row0 + r;
            const bool row_is_centre = py >= b && py < b + c;
            uint32_t texels[kPer];
            bool keep[kPer];
#pragma unroll
            for (uint32_t k = 0; k < kPer; k++) {
                keep[k] = false;
                texels[k] = 0u;  // split.wgsl:19-21: border pixels are zero until stitch fills them
                if (row_is_centre && col_is_centre[k]) {
                    const Axis ay = s_ay[r];
                    const uint2 t0 = split_fetch<FORMAT>(raster, ax[k], ay.i0), t1 = split_fetch<FORMAT>(raster, ax[k], ay.i1);
                    const HRow<FORMAT> top = split_hrow<FORMAT>(t0, ax[k].fr), bot = split_hrow<FORMAT>(t1, ax[k].fr);
                    keep[k] = !(top.valid && bot.valid);
                    texels[k] = split_texel<FORMAT>(top, bot, ay.fr);
                }
            }
            store(py, ex, texels, keep);
        }
    }
}

// ------------------------------------------------------------------------------------- downsample

template <uint32_t FORMAT>
__device__ __forceinline__ uint32_t downsample_texel(const typename Texel<FORMAT>::type* __restrict__ child,
                                                     uint32_t Tsz, uint32_t cx, uint32_t cy) {
    // OFFSETS (0,0),(0,1),(1,0),(1,1) as (dx,dy): downsample.wgsl:25
    uint32_t t[4];
    if (child) {
        t[0] = child[uint64_t(cy) * Tsz + cx];
        t[1] = child[uint64_t(cy + 1) * Tsz + cx];
        t[2] = child[uint64_t(cy) * Tsz + cx + 1];
        t[3] = child[uint64_t(cy + 1) * Tsz + cx + 1];
    } else {
        t[0] = t[1] = t[2] = t[3] = 0;  // child tile absent: the layer reads as zero
    }
    if constexpr (FORMAT == BT_FORMAT_R16) {
        float value = 0.0f, count = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (t[i] != 0) {  // any(child_value.xyz != 0) with xyz = (r, 0, 0)
                value += unorm16_to_float(t[i]);
                count += 1.0f;
            }
        if (count == 0.0f) return 0;  // 0/0: defined as "no data" (oracle)
        return float_to_unorm(value / count, 65535.0f);
    } else {
        float value[4] = {0.0f, 0.0f, 0.0f, 0.0f}, count = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if ((t[i] & 0x00FFFFFFu) != 0) {  // rgb != 0, alpha ignored
#pragma unroll
                for (int k = 0; k < 4; k++) value[k] += unorm8_to_float((t[i] >> (8 * k)) & 0xFFu);
                count += 1.0f;
            }
        if (count == 0.0f) return 0;
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) out |= float_to_unorm(value[k] / count, 255.0f) << (8 * k);
        return out;
    }
}

template <uint32_t FORMAT>
__global__ __launch_bounds__(256) void downsample_kernel(AttachmentMeta m, void* __restrict__ atlas,
                                                         const TaskDev* __restrict__ tasks, uint32_t row_blocks, uint32_t rows_per_block) {
    using T = typename Texel<FORMAT>::type;
    constexpr uint32_t kPer = Texel<FORMAT>::kPerEntry;
    const uint32_t task_index = blockIdx.x / row_blocks;
    const uint32_t row0 = (blockIdx.x % row_blocks) * rows_per_block;
    const TaskDev task = tasks[task_index];
    const uint32_t Tsz = m.texture_size, b = m.border_size, c = m.center_size;
    const uint32_t entries_per_row = Tsz / kPer;
    const uint32_t child_size = c / 2u;
    T* base = (T*)atlas;
    T* tile = base + uint64_t(task.atlas_index) * Tsz * Tsz;

    const uint32_t rows = min(rows_per_block, Tsz - row0);
    for (uint32_t e = threadIdx.x; e < entries_per_row * rows; e += blockDim.x) {
        const uint32_t py = row0 + e / entries_per_row;
        const uint32_t ex = e % entries_per_row;
        uint32_t texels[kPer];
#pragma unroll
        for (uint32_t k = 0; k < kPer; k++) {
            const uint32_t px = ex * kPer + k;
            if (is_border(px, py, b, c)) {
                texels[k] = 0;
                continue;
            }
            const uint32_t tx = px - b, ty = py - b;
            const uint32_t child_index = tx / child_size + 2u * (ty / child_size);
            const uint32_t layer = task.rel_index[child_index & 3u];
            const T* child = layer < m.atlas_size ? base + uint64_t(layer) * Tsz * Tsz : nullptr;
            texels[k] = downsample_texel<FORMAT>(child, Tsz, 2u * (tx % child_size) + b, 2u * (ty % child_size) + b);
        }
        if (py >= m.row_limit) continue;
        if constexpr (FORMAT == BT_FORMAT_R16)
            ((uint32_t*)tile)[(uint64_t(py) * Tsz) / 2 + ex] = texels[0] | (texels[1] << 16);
        else
            tile[uint64_t(py) * Tsz + ex] = texels[0];
    }
}

// ----------------------------------------------------------------------------------------- stitch

#include "bt_stitch.hpp"  // project_to_side, stitch_source, stitch_region_body

// one thread per apron pixel: 2*b*T (top+bottom rows) + 2*b*c (left+right columns) per tile
template <typename T>
__global__ __launch_bounds__(256) void stitch_kernel(AttachmentMeta m, void* __restrict__ atlas,
                                                     const TaskDev* __restrict__ tasks, uint32_t blocks_per_tile) {
    const uint32_t task_index = blockIdx.x / blocks_per_tile;
    const uint32_t i = (blockIdx.x % blocks_per_tile) * blockDim.x + threadIdx.x;
    const uint32_t Tsz = m.texture_size, b = m.border_size, c = m.center_size;
    const uint32_t n_rows = 2u * b * Tsz;
    if (i >= n_rows + 2u * b * c) return;
    const TaskDev task = tasks[task_index];
    uint32_t px, py;
    if (i < n_rows) {
        const uint32_t r = i / Tsz;
        px = i % Tsz;
        py = r < b ? r : (c + r);  // rows 0..b-1 and b+c..T-1
    } else {
        const uint32_t j = i - n_rows;
        const uint32_t k = j % (2u * b);
        py = b + j / (2u * b);
        px = k < b ? k : (c + k);
    }
    if (py >= m.row_limit) return;
    if (task.regions) {
        const uint32_t o = b + c;
        const int rx = px < b ? -1 : (px >= o ? 1 : 0), ry = py < b ? -1 : (py >= o ? 1 : 0);
        const uint32_t region = ry < 0 ? (rx < 0 ? 4u : (rx > 0 ? 5u : 0u)) : (ry > 0 ? (rx < 0 ? 7u : (rx > 0 ? 6u : 2u)) : (rx > 0 ? 1u : 3u));
        if (!((task.regions >> region) & 1u)) return;
    }
    uint32_t layer, sx, sy;
    stitch_source(task, px, py, Tsz, b, c, layer, sx, sy);
    T* base = (T*)atlas;
    T v = 0;
    if (layer < m.atlas_size && sx < Tsz && sy < Tsz) v = base[uint64_t(layer) * Tsz * Tsz + uint64_t(sy) * Tsz + sx];
    base[uint64_t(task.atlas_index) * Tsz * Tsz + uint64_t(py) * Tsz + px] = v;
}

// A workgroup = ONE apron region of one tile (bt_stitch.hpp)
template <typename T, uint32_t kPack>
__global__ __launch_bounds__(256) void stitch_region_kernel(AttachmentMeta m, void* __restrict__ atlas_, const TaskDev* __restrict__ tasks) {
    stitch_region_body<T, kPack>(m, atlas_, tasks[blockIdx.x]);
}

// R16 with an even border: one thread per apron pixel PAIR (a pair never straddles two regions) — half the
// stores, each 4 bytes; the left / right columns are isolated 4-byte accesses either way
__global__ __launch_bounds__(256) void stitch_pairs_kernel(AttachmentMeta m, uint16_t* __restrict__ atlas,
                                                           const TaskDev* __restrict__ tasks, uint32_t blocks_per_tile,
                                                           uint32_t pairs) {
    const uint32_t task_index = blockIdx.x / blocks_per_tile;
    const uint32_t i = (blockIdx.x % blocks_per_tile) * blockDim.x + threadIdx.x;
    const uint32_t Tsz = m.texture_size, b = m.border_size, c = m.center_size, o = b + c;
    const uint32_t row_pairs = b * Tsz;  // 2b rows of T / 2 pairs
    if (i >= pairs) return;              // all: + c rows of 2 * (b / 2) pairs; rows only: row_pairs
    const TaskDev task = tasks[task_index];
    uint32_t px, py;
    if (i < row_pairs) {
        const uint32_t r = i / (Tsz / 2u);
        px = 2u * (i % (Tsz / 2u));
        py = r < b ? r : (c + r);
    } else {
        const uint32_t j = i - row_pairs, k = j % b, d = k % (b / 2u);
        py = b + j / b;
        px = k < b / 2u ? 2u * d : o + 2u * d;
    }
    if (py >= m.row_limit) return;
    if (task.regions) {  // a pair lies in one region (b even)
        const int rx = px < b ? -1 : (px >= o ? 1 : 0), ry = py < b ? -1 : (py >= o ? 1 : 0);
        const uint32_t region = ry < 0 ? (rx < 0 ? 4u : (rx > 0 ? 5u : 0u)) : (ry > 0 ? (rx < 0 ? 7u : (rx > 0 ? 6u : 2u)) : (rx > 0 ? 1u : 3u));
        if (!((task.regions >> region) & 1u)) return;
    }
    uint32_t v[2];
#pragma unroll
    for (uint32_t e = 0; e < 2; e++) {
        uint32_t layer, sx, sy;
        stitch_source(task, px + e, py, Tsz, b, c, layer, sx, sy);
        v[e] = (layer < m.atlas_size && sx < Tsz && sy < Tsz) ? atlas[uint64_t(layer) * Tsz * Tsz + uint64_t(sy) * Tsz + sx] : 0u;
    }
    *reinterpret_cast<uint32_t*>(atlas + uint64_t(task.atlas_index) * Tsz * Tsz + uint64_t(py) * Tsz + px) = v[0] | (v[1] << 16);
}

// -------------------------------------------------------------------------------------- mip chain

// AttachmentData::generate_mipmaps: R16 = truncating mean of the non-zero texels, RGBA8 = sum / 4
__global__ __launch_bounds__(256) void mip_r16_kernel(const uint16_t* __restrict__ parent, uint16_t* __restrict__ child,
                                                      uint32_t parent_size, uint32_t layers) {
    const uint32_t cs = parent_size >> 1;
    const uint64_t total = uint64_t(cs) * cs * layers;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t cx = uint32_t(i % cs), cy = uint32_t((i / cs) % cs);
        const uint64_t layer = i / (uint64_t(cs) * cs);
        const uint16_t* p = parent + layer * parent_size * parent_size + uint64_t(2 * cy) * parent_size + 2 * cx;
        const uint32_t v[4] = {p[0], p[parent_size], p[1], p[parent_size + 1]};
        uint32_t value = 0, count = 0;
#pragma unroll
        for (int k = 0; k < 4; k++)
            if (v[k] != 0) {
                value += v[k];
                count++;
            }
        child[i] = count == 0 ? uint16_t(0) : uint16_t(value / count);
    }
}

__global__ __launch_bounds__(256) void mip_rgba8_kernel(const uint32_t* __restrict__ parent, uint32_t* __restrict__ child,
                                                        uint32_t parent_size, uint32_t layers) {
    const uint32_t cs = parent_size >> 1;
    const uint64_t total = uint64_t(cs) * cs * layers;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t cx = uint32_t(i % cs), cy = uint32_t((i / cs) % cs);
        const uint64_t layer = i / (uint64_t(cs) * cs);
        const uint32_t* p = parent + layer * parent_size * parent_size + uint64_t(2 * cy) * parent_size + 2 * cx;
        const uint32_t v[4] = {p[0], p[1], p[parent_size], p[parent_size + 1]};
        uint32_t out = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const uint32_t sh = 8 * k;
            const uint32_t sum = ((v[0] >> sh) & 0xFFu) + ((v[1] >> sh) & 0xFFu) + ((v[2] >> sh) & 0xFFu) + ((v[3] >> sh) & 0xFFu);
            out |= (sum / 4u) << sh;
        }
        child[i] = out;
    }
}

// -------------------------------------------------------------------------- atlas sampling (CPU-side queries)

// AtlasAttachment::sample + AttachmentData::sample (tile_atlas.rs:249-258, terrain_data/mod.rs:220-263), one lookup
// per thread.  Same operation order as the reference's f32 code (glam lerp = a + (b - a) * s, no contraction);
// texel coordinates clamped into the tile where the reference would index out of bounds.
__global__ __launch_bounds__(256) void sample_kernel(AttachmentMeta m, const void* __restrict__ atlas, const bt_tile_lookup* __restrict__ lookups,
                                                     uint32_t count, float4* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const bt_tile_lookup l = lookups[i];
    if (l.atlas_index >= m.atlas_size) {  // INVALID_ATLAS_INDEX (or out of range): "Todo: Handle this better" -> zero
        out[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        return;
    }
    const uint32_t T = m.texture_size;
    const float scale = float(m.center_size) / float(T), offset = float(m.border_size) / float(T);
    float rem[2];
    int ixy[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const float u = l.atlas_uv[a] * scale + offset;
        const float uv = u * float(T) - 0.5f;
        rem[a] = fmodf(uv, 1.0f);
        ixy[a] = int(uv);
    }
    float v[2][2][4];
#pragma unroll
    for (int x = 0; x < 2; x++)
#pragma unroll
        for (int y = 0; y < 2; y++) {
            const uint32_t px = uint32_t(min(max(ixy[0] + x, 0), int(T) - 1)), py = uint32_t(min(max(ixy[1] + y, 0), int(T) - 1));
            const uint64_t index = uint64_t(l.atlas_index) * T * T + uint64_t(py) * T + px;
            if (m.format == BT_FORMAT_R16) {
                v[x][y][0] = unorm16_to_float(((const uint16_t*)atlas)[index]);
                v[x][y][1] = v[x][y][2] = v[x][y][3] = 0.0f;
            } else {
                const uint32_t t = ((const uint32_t*)atlas)[index];
#pragma unroll
                for (int k = 0; k < 4; k++) v[x][y][k] = unorm8_to_float((t >> (8 * k)) & 0xFFu);
            }
        }
    float r[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float a = v[0][0][k] + (v[0][1][k] - v[0][0][k]) * rem[1];
        const float b = v[1][0][k] + (v[1][1][k] - v[1][0][k]) * rem[1];
        r[k] = a + (b - a) * rem[0];
    }
    out[i] = make_float4(r[0], r[1], r[2], r[3]);
}

// -------------------------------------------------------------------- synthetic fBm (bench input)

__device__ __forceinline__ uint64_t hash2(uint64_t ix, uint64_t iy, uint32_t seed) {
    uint64_t h = (ix * 0x85EBCA6Bull + iy * 0xC2B2AE35ull + seed) & 0xFFFFFFFFull;
    h ^= h >> 15;
    h = (h * 0x2C1B3C6Dull) & 0xFFFFFFFFull;
    h ^= h >> 12;
    h = (h * 0x297A2D39ull) & 0xFFFFFFFFull;
    h ^= h >> 15;
    return h & 0xFFFFull;
}

__device__ __forceinline__ uint64_t value_noise(uint64_t x, uint64_t y, uint64_t cell, uint32_t seed) {
    const uint64_t ix = x / cell, fx = x % cell, iy = y / cell, fy = y % cell;
    const uint64_t wx = (fx * 4096ull) / cell, wy = (fy * 4096ull) / cell;
    const uint64_t v00 = hash2(ix, iy, seed), v10 = hash2(ix + 1, iy, seed);
    const uint64_t v01 = hash2(ix, iy + 1, seed), v11 = hash2(ix + 1, iy + 1, seed);
    const uint64_t top = v00 * (4096ull - wx) + v10 * wx;
    const uint64_t bot = v01 * (4096ull - wx) + v11 * wx;
    return (top * (4096ull - wy) + bot * wy) >> 24;
}

__global__ __launch_bounds__(256) void synth_fbm_kernel(uint8_t* __restrict__ dst, uint32_t w, uint32_t h, uint64_t pitch,
                                                        uint32_t x0, uint32_t y0, uint32_t base_cell, uint32_t octaves,
                                                        uint32_t seed) {
    const uint64_t total = uint64_t(w) * h;
    for (uint64_t i = uint64_t(blockIdx.x) * blockDim.x + threadIdx.x; i < total; i += uint64_t(gridDim.x) * blockDim.x) {
        const uint32_t x = uint32_t(i % w), y = uint32_t(i / w);
        uint64_t sum = 0, amp_total = 0, cell = base_cell;
        for (uint32_t o = 0; o < octaves; o++) {
            if (cell < 1) cell = 1;
            const uint64_t amp = 1ull << (octaves - 1 - o);
            sum += value_noise(uint64_t(x) + x0, uint64_t(y) + y0, cell, seed + 0x9E3779B9u * (o + 1)) * amp;
            amp_total += amp;
            cell /= 2;
        }
        const uint64_t v = sum / amp_total;
        ((uint16_t*)(dst + uint64_t(y) * pitch))[x] = uint16_t(1ull + (v * 65534ull) / 65535ull);
    }
}

}  // namespace

// ------------------------------------------------------------------------------------- launchers

static bt_status check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, what);
    return BT_OK;
}

bt_status launch_split(bt_ctx* ctx, const AttachmentMeta& m, void* atlas, const TaskDev* tasks, uint32_t n,
                       const RasterDev* rasters) {
    if (!n) return BT_OK;
    const uint32_t row_blocks = (m.texture_size + kRows - 1) / kRows;
    if (m.format == BT_FORMAT_R16)
        split_kernel<BT_FORMAT_R16><<<n * row_blocks, 256, 0, ctx->stream>>>(m, atlas, tasks, rasters, row_blocks);
    else
        split_kernel<BT_FORMAT_RGBA8><<<n * row_blocks, 256, 0, ctx->stream>>>(m, atlas, tasks, rasters, row_blocks);
    return check_launch("split_kernel");
}

bt_status launch_downsample(bt_ctx* ctx, const AttachmentMeta& m, void* atlas, const TaskDev* tasks, uint32_t n) {
    if (!n) return BT_OK;
    const uint32_t rows_per_block = n <= 4 ? 1u : (n <= 16 ? 2u : kRows);
    const uint32_t row_blocks = (m.texture_size + rows_per_block - 1) / rows_per_block;
    if (m.format == BT_FORMAT_R16)
        downsample_kernel<BT_FORMAT_R16><<<n * row_blocks, 256, 0, ctx->stream>>>(m, atlas, tasks, row_blocks, rows_per_block);
    else
        downsample_kernel<BT_FORMAT_RGBA8><<<n * row_blocks, 256, 0, ctx->stream>>>(m, atlas, tasks, row_blocks, rows_per_block);
    return check_launch("downsample_kernel");
}

bt_status launch_stitch(bt_ctx* ctx, const AttachmentMeta& m, void* atlas, const TaskDev* tasks, uint32_t n, bool rows_only, bool one_region) {
    if (!n || m.border_size == 0) return BT_OK;
    if (one_region) {
        if (m.format == BT_FORMAT_R16 && m.border_size % 2u == 0 && m.texture_size % 2u == 0) stitch_region_kernel<uint16_t, 2><<<n, 256, 0, ctx->stream>>>(m, atlas, tasks);
        else if (m.format == BT_FORMAT_R16) stitch_region_kernel<uint16_t, 1><<<n, 256, 0, ctx->stream>>>(m, atlas, tasks);
        else stitch_region_kernel<uint32_t, 1><<<n, 256, 0, ctx->stream>>>(m, atlas, tasks);
        return check_launch("stitch_region_kernel");
    }
    const uint32_t apron = 2u * m.border_size * (m.texture_size + m.center_size);
    const uint32_t blocks = (apron + 255u) / 256u;
    if (m.format == BT_FORMAT_R16 && m.border_size % 2u == 0 && m.texture_size % 2u == 0) {
        const uint32_t pairs = rows_only ? m.border_size * m.texture_size : apron / 2u;
        const uint32_t pair_blocks = (pairs + 255u) / 256u;
        stitch_pairs_kernel<<<n * pair_blocks, 256, 0, ctx->stream>>>(m, (uint16_t*)atlas, tasks, pair_blocks, pairs);
    } else if (m.format == BT_FORMAT_R16)
        stitch_kernel<uint16_t><<<n * blocks, 256, 0, ctx->stream>>>(m, atlas, tasks, blocks);
    else
        stitch_kernel<uint32_t><<<n * blocks, 256, 0, ctx->stream>>>(m, atlas, tasks, blocks);
    return check_launch("stitch_kernel");
}

bt_status launch_sample(bt_ctx* ctx, const AttachmentMeta& m, const void* atlas, const bt_tile_lookup* lookups, uint32_t count, float* out) {
    if (!count) return BT_OK;
    sample_kernel<<<(count + 255u) / 256u, 256, 0, ctx->stream>>>(m, atlas, lookups, count, (float4*)out);
    return check_launch("sample_kernel");
}

bt_status launch_mip_level(bt_ctx* ctx, uint32_t format, const void* parent, void* child, uint32_t parent_size,
                           uint32_t layers) {
    const uint64_t total = uint64_t(parent_size >> 1) * (parent_size >> 1) * layers;
    if (!total) return BT_OK;
    const uint32_t grid = uint32_t(std::min<uint64_t>((total + 255) / 256, 256ull * 16));
    if (format == BT_FORMAT_R16)
        mip_r16_kernel<<<grid, 256, 0, ctx->stream>>>((const uint16_t*)parent, (uint16_t*)child, parent_size, layers);
    else
        mip_rgba8_kernel<<<grid, 256, 0, ctx->stream>>>((const uint32_t*)parent, (uint32_t*)child, parent_size, layers);
    return check_launch("mip kernel");
}

// Tile download as ONE kernel per chunk: the layers of a chunk gathered straight into a pinned host buffer (16-byte vectors, the GPU writes
// host memory over PCIe) — instead of one copy-engine call per run of consecutive layers.  tile_bytes % 16 == 0 (rows are whole dwords, T even).
struct GatherLayers {
    uint32_t layer[64];
};
typedef uint32_t gather_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void gather_layers_kernel(const gather_u32x4* __restrict__ atlas, GatherLayers list, gather_u32x4* __restrict__ dst, uint32_t vec_per_tile) {
    const uint32_t tile = blockIdx.y;
    const gather_u32x4* src = atlas + uint64_t(list.layer[tile]) * vec_per_tile;
    gather_u32x4* out = dst + uint64_t(tile) * vec_per_tile;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < vec_per_tile; i += gridDim.x * 256u) out[i] = __builtin_nontemporal_load(src + i);
}
bt_status launch_gather_layers(hipStream_t stream, const void* atlas, const uint32_t* layers, uint32_t count, void* pinned_dst, uint64_t tile_bytes) {
    if (!count) return BT_OK;
    if (count > 64 || tile_bytes % 16u) {
        set_error("gather of %u layers of %llu bytes", count, (unsigned long long)tile_bytes);
        return BT_ERR_INVALID_ARGUMENT;
    }
    GatherLayers list{};
    for (uint32_t i = 0; i < count; i++) list.layer[i] = layers[i];
    const uint32_t vec = uint32_t(tile_bytes / 16u);
    const uint32_t bx = std::max(1u, std::min(32u, vec / 1024u));  // >= 4 vectors per thread
    gather_layers_kernel<<<dim3(bx, count), 256, 0, stream>>>((const gather_u32x4*)atlas, list, (gather_u32x4*)pinned_dst, vec);
    return check_launch("gather_layers_kernel");
}

bt_status launch_synth_fbm(bt_ctx* ctx, void* dst, uint32_t w, uint32_t h, uint64_t pitch, uint32_t x0, uint32_t y0,
                           uint32_t base_cell, uint32_t octaves, uint32_t seed) {
    const uint64_t total = uint64_t(w) * h;
    const uint32_t grid = uint32_t(std::min<uint64_t>((total + 255) / 256, 256ull * 32));
    synth_fbm_kernel<<<grid, 256, 0, ctx->stream>>>((uint8_t*)dst, w, h, pitch, x0, y0, base_cell, octaves, seed);
    return check_launch("synth_fbm_kernel");
}

}  // namespace bt
