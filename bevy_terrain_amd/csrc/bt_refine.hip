// Tiling prepass (UDLOD tile refinement) as ONE persistent launch.
//
// Reference: TilingPrepassNode::run (render/tiling_prepass.rs:204-272) records 2*refinement_count+3
// dependent dispatches per view per frame — prepare_root, then (refine_tiles indirect, prepare_next) x
// refinement_count, refine_tiles, prepare_render — with two global atomics per tile
// (shaders/tiling_prepass/refine_tiles.wgsl:5-15) and "very low" occupancy (docs/implementation.md:62-65).
//
// Here one 1024-thread workgroup (16 wavefronts of 64) runs the whole schedule: the per-pass state
// (`Parameters`, types.wgsl:43-48) lives in LDS, passes are separated by __syncthreads() instead of
// kernel boundaries, and the child / final appends are a stable compaction — wave64 __ballot +
// __popcll ranks inside a wave, a 16-entry LDS scan across waves — so there are no global atomics and
// the output order is exactly the order of the reference run with invocations taken in id order.
// The ping-pong layout of `temporary_tiles` (parents read from one end, children appended from the
// other, prepare_prepass.wgsl:25-36) is kept so the buffers are bit-compatible.
//
// Arithmetic contract: IEEE binary32, one rounding per written operation (-ffp-contract=off), same
// operation order as oracle/bt_oracle.c (functions.wgsl:14-29,73-96,117-188).
#include "bt_internal.hpp"

struct bt_tiling_prepass {
    bt_ctx* ctx = nullptr;
    uint32_t capacity = 0;
    bt_tile_coordinate* temporary_tiles = nullptr;
    bt_tile_coordinate* final_tiles = nullptr;
    bt_indirect* indirect = nullptr;
    uint32_t* counters = nullptr;  // [0] final count, [1] overflow flag, [2] tiles visited, [3] passes
};

namespace bt {
namespace {

constexpr uint32_t kThreads = 1024;
constexpr uint32_t kWaves = kThreads / 64;

struct Coordinate {  // types.wgsl:31-40
    uint32_t side, lod, x, y;
    float u, v;
};

// functions.wgsl:164-188; pow(2.0, f32(d)) is exact (ldexpf)
__device__ __forceinline__ void coordinate_change_lod(Coordinate& c, uint32_t new_lod) {
    const int d = int(new_lod) - int(c.lod);
    if (d == 0) return;
    const uint32_t delta_count = 1u << uint32_t(d < 0 ? -d : d);
    const float delta_size = ldexpf(1.0f, d);
    c.lod = new_lod;
    if (d > 0) {
        const float su = c.u * delta_size, sv = c.v * delta_size;
        c.x = c.x * delta_count + uint32_t(su);
        c.y = c.y * delta_count + uint32_t(sv);
        c.u = su - truncf(su);
        c.v = sv - truncf(sv);
    } else {
        const uint32_t x = c.x, y = c.y, sh = uint32_t(-d);  // delta_count = 2^sh: quotient and remainder by shift and mask
        c.x = x >> sh;
        c.y = y >> sh;
        c.u = (float(x & (delta_count - 1u)) + c.u) * delta_size;
        c.v = (float(y & (delta_count - 1u)) + c.v) * delta_size;
    }
}

__device__ __forceinline__ float length3(float x, float y, float z) { return sqrtf(x * x + y * y + z * z); }

// refine_tiles.wgsl:17-22 -> compute_subdivision_coordinate (functions.wgsl:133-154) ->
// approximate_view_distance (:117-131) -> compute_local_position (:73-96)
__device__ bool should_be_divided(const bt_view_state& v, const bt_tile_coordinate& tile) {
    Coordinate vc{tile.side, v.origin_lod, uint32_t(v.sides[tile.side].view_xy[0]), uint32_t(v.sides[tile.side].view_xy[1]),
                  v.sides[tile.side].view_uv[0], v.sides[tile.side].view_uv[1]};
    coordinate_change_lod(vc, tile.lod);
    const int off_x = int(vc.x) - int(tile.x), off_y = int(vc.y) - int(tile.y);
    const float uv_x = off_x < 0 ? 0.0f : (off_x > 0 ? 1.0f : vc.u);
    const float uv_y = off_y < 0 ? 0.0f : (off_y > 0 ? 1.0f : vc.v);

    // tile_count(lod) = 2^lod: x / 2^lod == x * 2^-lod bit for bit (no underflow at these magnitudes), and so is / 0.5 == * 2 —
    // five of the function's eleven IEEE divisions (each ~10 instructions, and this kernel is one CU's VALU)
    const float inv_tc = __builtin_bit_cast(float, (127u - tile.lod) << 23);
    float u = (float(tile.x) + uv_x) * inv_tc;
    float w = (float(tile.y) + uv_y) * inv_tc;
    float lx, ly, lz;
    if (v.spherical) {
        const float C_SQR = 0.87f * 0.87f;
        u = (u - 0.5f) * 2.0f;
        w = (w - 0.5f) * 2.0f;
        u = u / sqrtf(1.0f + C_SQR - C_SQR * u * u);
        w = w / sqrtf(1.0f + C_SQR - C_SQR * w * w);
        switch (tile.side) {
            case 0: lx = -1.0f; ly = -w; lz = u; break;
            case 1: lx = u; ly = -w; lz = 1.0f; break;
            case 2: lx = u; ly = 1.0f; lz = w; break;
            case 3: lx = 1.0f; ly = -u; lz = w; break;
            case 4: lx = w; ly = -u; lz = -1.0f; break;
            case 5: lx = w; ly = -1.0f; lz = u; break;
            default: lx = ly = lz = 0.0f; break;
        }
        const float l = length3(lx, ly, lz);
        lx = lx / l;
        ly = ly / l;
        lz = lz / l;
    } else {
        lx = u - 0.5f;
        ly = 0.0f;
        lz = w - 0.5f;
    }
    const float* m = v.world_from_local;  // 3 columns + translation
    const float wx = (m[0] * lx + m[3] * ly + m[6] * lz) + m[9];
    const float wy = (m[1] * lx + m[4] * ly + m[7] * lz) + m[10];
    const float wz = (m[2] * lx + m[5] * ly + m[8] * lz) + m[11];
    const float nx0 = v.spherical ? lx : 0.0f, ny0 = v.spherical ? ly : 1.0f, nz0 = v.spherical ? lz : 0.0f;
    const float* t = v.local_from_world_transpose;
    float nx = t[0] * nx0 + t[3] * ny0 + t[6] * nz0;
    float ny = t[1] * nx0 + t[4] * ny0 + t[7] * nz0;
    float nz = t[2] * nx0 + t[5] * ny0 + t[8] * nz0;
    const float nl = length3(nx, ny, nz);
    nx = nx / nl;
    ny = ny / nl;
    nz = nz / nl;
    const float dx = (wx + v.approximate_height * nx) - v.world_position[0];
    const float dy = (wy + v.approximate_height * ny) - v.world_position[1];
    const float dz = (wz + v.approximate_height * nz) - v.world_position[2];
    const float view_distance = length3(dx, dy, dz);
    return view_distance < v.subdivision_distance * inv_tc;
}

__global__ __launch_bounds__(kThreads) void tiling_prepass_kernel(bt_view_state view, uint32_t capacity,
                                                                  bt_tile_coordinate* __restrict__ temporary_tiles,
                                                                  bt_tile_coordinate* __restrict__ final_tiles,
                                                                  bt_indirect* __restrict__ indirect,
                                                                  uint32_t* __restrict__ counters) {
    // The pass state (Parameters, types.wgsl:43-48) is uniform and kept in registers by every thread; only the
    // per-wave counts of a sweep go through LDS (double-buffered by sweep parity: ONE barrier per sweep).
    __shared__ uint32_t s_divide[2][kWaves], s_final[2][kWaves];

    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const int N = int(capacity);

    // prepare_root (prepare_prepass.wgsl:4-23)
    int counter = -1, child_index = N - 1, final_index = 0;
    uint32_t tile_count = view.spherical ? 6u : 1u, visited = 0, sweep = 0;
    bool overflow = false;
    if (tid < tile_count) temporary_tiles[tid] = {tid, 0u, 0u, 0u};
    __syncthreads();

    for (uint32_t pass = 0; pass <= view.refinement_count; pass++) {
        // refine_tiles (refine_tiles.wgsl:33-44), 1024 invocation ids per sweep
        for (uint32_t base = 0; base < tile_count; base += kThreads, sweep++) {
            const uint32_t id = base + tid;
            const bool active = id < tile_count;
            bt_tile_coordinate tile{};
            bool divide = false;
            if (active) {
                const int parent_index = (N - 1) * (counter > 0 ? 1 : 0) - int(id) * counter;  // :9-11
                tile = temporary_tiles[parent_index];
                divide = should_be_divided(view, tile);
            }
            const bool fin = active && !divide;
            const unsigned long long ballot_d = __ballot(divide), ballot_f = __ballot(fin);
            if (lane == 0) {
                s_divide[sweep & 1u][wave] = uint32_t(__popcll(ballot_d));
                s_final[sweep & 1u][wave] = uint32_t(__popcll(ballot_f));
            }
            __syncthreads();
            uint32_t before_d = 0, before_f = 0, total_d = 0, total_f = 0;
#pragma unroll
            for (uint32_t w = 0; w < kWaves; w++) {
                const uint32_t d = s_divide[sweep & 1u][w], f = s_final[sweep & 1u][w];
                before_d += w < wave ? d : 0u;
                before_f += w < wave ? f : 0u;
                total_d += d;
                total_f += f;
            }
            const unsigned long long below = (1ull << lane) - 1ull;
            if (divide) {  // subdivide (:24-31): 4 children at consecutive child_index() values
                const int rank = int(before_d + uint32_t(__popcll(ballot_d & below)));
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const int ci = child_index + counter * (4 * rank + int(i));
                    if (ci >= 0 && ci < N)
                        temporary_tiles[ci] = {tile.side, tile.lod + 1u, (tile.x << 1) + (i & 1u), (tile.y << 1) + ((i >> 1) & 1u)};
                }
            }
            if (fin) {
                const int fi = final_index + int(before_f + uint32_t(__popcll(ballot_f & below)));
                if (fi < N) final_tiles[fi] = tile;
            }
            child_index += counter * 4 * int(total_d);
            final_index += int(total_f);
            visited += min(kThreads, tile_count - base);
            // children may not run into the parents still to be read, nor finals past the buffer
            const int children_so_far = counter > 0 ? child_index : (N - 1 - child_index);
            if (children_so_far + int(tile_count) > N || final_index > N) overflow = true;
            if (overflow) break;  // uniform: the buffers are too small, stop before indices run wild
        }
        if (overflow || pass == view.refinement_count) break;
        // prepare_next (prepare_prepass.wgsl:25-36)
        if (counter == 1) {
            tile_count = uint32_t(child_index);
            child_index = N - 1;
        } else {
            tile_count = uint32_t(N - 1 - child_index);
            child_index = 0;
        }
        counter = -counter;
        if (tile_count == 0) break;  // nothing left to refine: the remaining passes of the reference are no-ops
        __syncthreads();  // orders this pass's child stores before the next pass's parent loads
    }

    // prepare_render (prepare_prepass.wgsl:38-44)
    if (tid == 0) {
        *indirect = {view.vertices_per_tile * uint32_t(final_index), 1u, 0u, 0u};
        counters[0] = uint32_t(final_index);
        counters[1] = overflow ? 1u : 0u;
        counters[2] = visited;
        counters[3] = view.refinement_count + 1;
    }
}

}  // namespace
}  // namespace bt

using namespace bt;

extern "C" {

bt_status bt_tiling_prepass_create(bt_ctx* ctx, uint32_t geometry_tile_count, bt_tiling_prepass** out) {
    if (!ctx || !out || geometry_tile_count < 8) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    bt_tiling_prepass* t = new bt_tiling_prepass();
    t->ctx = ctx;
    t->capacity = geometry_tile_count;
    // TerrainViewData::new: two buffers of geometry_tile_count TileCoordinates (terrain_view_bind_group.rs:130-142)
    hipError_t e = hipMalloc((void**)&t->temporary_tiles, sizeof(bt_tile_coordinate) * size_t(geometry_tile_count));
    if (e == hipSuccess) e = hipMalloc((void**)&t->final_tiles, sizeof(bt_tile_coordinate) * size_t(geometry_tile_count));
    if (e == hipSuccess) e = hipMalloc((void**)&t->indirect, sizeof(bt_indirect));
    if (e == hipSuccess) e = hipMalloc((void**)&t->counters, 16 * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(t->counters, 0, 16 * sizeof(uint32_t), ctx->stream);
    if (e != hipSuccess) {
        bt_tiling_prepass_destroy(t);
        return hip_fail(e, "tiling prepass buffers");
    }
    *out = t;
    return BT_OK;
}

void bt_tiling_prepass_destroy(bt_tiling_prepass* t) {
    if (!t) return;
    hipSetDevice(t->ctx->device);
    if (t->temporary_tiles) hipFree(t->temporary_tiles);
    if (t->final_tiles) hipFree(t->final_tiles);
    if (t->indirect) hipFree(t->indirect);
    if (t->counters) hipFree(t->counters);
    delete t;
}

bt_status bt_tiling_prepass_run(bt_tiling_prepass* t, const bt_view_state* view) {
    if (!t || !view) return BT_ERR_INVALID_ARGUMENT;
    if (view->refinement_count > 31) {
        set_error("refinement_count %u > 31 (tile x/y are u32)", view->refinement_count);
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (view->origin_lod > 31) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(t->ctx->device));
    const uint32_t capacity = std::min(t->capacity, view->geometry_tile_count ? view->geometry_tile_count : t->capacity);
    tiling_prepass_kernel<<<1, kThreads, 0, t->ctx->stream>>>(*view, capacity, t->temporary_tiles, t->final_tiles, t->indirect,
                                                               t->counters);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "tiling_prepass_kernel");
    return BT_OK;
}

bt_status bt_tiling_prepass_buffers(const bt_tiling_prepass* t, void** final_tiles, void** indirect) {
    if (!t) return BT_ERR_INVALID_ARGUMENT;
    if (final_tiles) *final_tiles = t->final_tiles;
    if (indirect) *indirect = t->indirect;
    return BT_OK;
}

bt_status bt_tiling_prepass_read(bt_tiling_prepass* t, bt_tile_coordinate* out, uint32_t cap, uint32_t* count, bt_indirect* indirect) {
    if (!t || !count) return BT_ERR_INVALID_ARGUMENT;
    uint32_t counters[4] = {0, 0, 0, 0};
    BT_HIP(hipMemcpyAsync(counters, t->counters, sizeof counters, hipMemcpyDeviceToHost, t->ctx->stream));
    if (indirect) BT_HIP(hipMemcpyAsync(indirect, t->indirect, sizeof(bt_indirect), hipMemcpyDeviceToHost, t->ctx->stream));
    BT_HIP(hipStreamSynchronize(t->ctx->stream));
    *count = counters[0];
    if (counters[1]) {
        set_error("tiling prepass overflowed its %u-entry tile buffers", t->capacity);
        return BT_ERR_OVERFLOW;
    }
    if (out && counters[0]) {
        const uint32_t n = std::min(cap, counters[0]);
        BT_HIP(hipMemcpyAsync(out, t->final_tiles, sizeof(bt_tile_coordinate) * size_t(n), hipMemcpyDeviceToHost, t->ctx->stream));
        BT_HIP(hipStreamSynchronize(t->ctx->stream));
    }
    return BT_OK;
}

}  // extern "C"
