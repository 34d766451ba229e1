// TileTree + GpuTileTree on the device.
//
// Reference, per frame and view, on the CPU (plugin.rs:46-56): TileTree::compute_requests -> update
// (terrain_data/tile_tree.rs:268-359: sides x lods x tree_size^2 nodes, f64 distance per node, request / release
// lists), TileAtlas::update (tile_atlas.rs:574-601), TileTree::adjust_to_tile_atlas (:363-374: get_best_tile per
// node), TileTree::approximate_height (:376-386 -> sample_height, terrain_data/mod.rs:265-307); then
// GpuTileTree::extract / prepare copy the node table and the origins to the GPU every frame (gpu_tile_tree.rs:71-95).
//
// Here the node tables live in HBM and never travel: `update` is ONE launch of one 1024-thread workgroup over all
// nodes (the released / requested lists come out in the reference's push order through a stable ballot + prefix-sum
// compaction — no atomics), `adjust` looks every node's best tile up in a device copy of the atlas's tile states (an
// open-addressing table, re-uploaded only when the states changed), sampling runs against the atlas in HBM.  The
// streaming state machine itself (LRU, file loads) stays on the host (bt_host.cpp), like the reference's.
//
// Arithmetic: IEEE binary64 / binary32, one rounding per written operation, see bt_model.hpp for the definitions
// where the reference defers to glam / libm.  compute_blend's log2 is the platform's (OCML here, libm in the oracle).
#include <cstring>

#include "bt_internal.hpp"
#include "bt_model.hpp"

using namespace bt;
using namespace bt::model;

namespace {

constexpr uint32_t kThreads = 1024;
constexpr uint32_t kWaves = kThreads / 64;
constexpr uint32_t kInvalid = 0xFFFFFFFFu;

struct NodeState {  // TileState of the tree (tile_tree.rs:28-43)
    bt_tile_coordinate coordinate;
    uint32_t requested;  // RequestState::Requested
};

struct TreeParams {
    Model model;
    uint32_t lod_count, tree_size, sides;
    double load_distance, blend_distance;
    float blend_range, approximate_height;
    V3 view_world_position;
    Coordinate view_coordinate[6];  // the view coordinate projected to every side
};

// One device copy of TileAtlasState::tile_states: open addressing, linear probing; key = coordinate
struct StateSlot {
    bt_tile_coordinate coordinate;  // side == kInvalid: empty
    uint32_t atlas_index;
    uint32_t loaded;
};

__host__ __device__ __forceinline__ uint32_t hash_coordinate(bt_tile_coordinate c) {
    uint64_t h = (uint64_t(c.side) << 59) ^ (uint64_t(c.lod) << 53) ^ (uint64_t(c.x) << 26) ^ uint64_t(c.y);
    h ^= h >> 31;
    h *= 0x9E3779B97F4A7C15ull;
    return uint32_t(h ^ (h >> 29));
}

// ---- update ---------------------------------------------------------------------------------------------------

// TileTree::update (tile_tree.rs:268-333).  Node n = ((side * lods + lod) * ts + x) * ts + y is the reference's loop
// order (iproduct!(0..ts, 0..ts): x outer); the node's table slot is [side][lod][tile.x % ts][tile.y % ts].
__global__ __launch_bounds__(kThreads) void tile_tree_update_kernel(TreeParams P, NodeState* __restrict__ nodes, uint32_t* __restrict__ origins,
                                                                    bt_tile_coordinate* __restrict__ released, bt_tile_coordinate* __restrict__ requested,
                                                                    uint32_t* __restrict__ counts, const float* __restrict__ height) {
    __shared__ uint32_t s_rel[2][kWaves], s_req[2][kWaves];
    // the tree's height lives on the device (bt_frame_update: the host's copy may lag a frame).  Kept beside P: writing into the
    // by-value kernel argument makes the compiler copy the struct to scratch
    const float approximate_height = height ? *height : P.approximate_height;
    const uint32_t tid = threadIdx.x, wave = tid >> 6, lane = tid & 63u;
    const uint32_t ts = P.tree_size, per_layer = ts * ts, total = P.sides * P.lod_count * per_layer;
    uint32_t rel_base = 0, req_base = 0, sweep = 0;
    for (uint32_t first = 0; first < total; first += kThreads, sweep++) {
        const uint32_t n = first + tid;
        bool push_rel = false, push_req = false;
        bt_tile_coordinate rel_c{}, req_c{};
        if (n < total) {
            const uint32_t layer = n / per_layer, in_layer = n - layer * per_layer;
            const uint32_t side = layer / P.lod_count, lod = layer - side * P.lod_count;
            const uint32_t x = in_layer / ts, y = in_layer - x * ts;
            const Coordinate vc = P.view_coordinate[side];
            uint32_t origin[2];
            compute_origin(vc, lod, ts, origin);
            if (in_layer == 0) {
                origins[2 * layer] = origin[0];
                origins[2 * layer + 1] = origin[1];
            }
            const bt_tile_coordinate tile = {side, lod, origin[0] + x, origin[1] + y};
            const double tile_distance = compute_tile_distance(tile, vc, P.model, approximate_height, P.view_world_position);
            const double load_distance = P.load_distance / double(1u << lod);
            const bool want = lod == 0 || tile_distance < load_distance;
            NodeState* slot = nodes + layer * per_layer + (tile.x % ts) * ts + (tile.y % ts);
            NodeState st = *slot;
            // the slot refers to a new tile: release the old one (:300-308)
            if (!(st.coordinate.side == tile.side && st.coordinate.lod == tile.lod && st.coordinate.x == tile.x && st.coordinate.y == tile.y)) {
                if (st.requested) {
                    st.requested = 0;
                    push_rel = true;
                    rel_c = st.coordinate;
                }
                st.coordinate = tile;
            }
            // request or release by distance (:311-321)
            if (!st.requested && want) {
                st.requested = 1;
                push_req = true;
                req_c = st.coordinate;
            } else if (st.requested && !want) {
                st.requested = 0;
                push_rel = true;
                rel_c = st.coordinate;
            }
            *slot = st;
        }
        // stable compaction in node order: rank inside the wave by ballot, across waves by a 16-entry scan
        const unsigned long long rel_bits = __ballot(push_rel), req_bits = __ballot(push_req);
        const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64u - lane));
        const uint32_t parity = sweep & 1u;
        if (lane == 0) {
            s_rel[parity][wave] = uint32_t(__popcll(rel_bits));
            s_req[parity][wave] = uint32_t(__popcll(req_bits));
        }
        __syncthreads();
        uint32_t rel_before = 0, req_before = 0, rel_total = 0, req_total = 0;
#pragma unroll
        for (uint32_t w = 0; w < kWaves; w++) {
            const uint32_t a = s_rel[parity][w], b = s_req[parity][w];
            if (w < wave) {
                rel_before += a;
                req_before += b;
            }
            rel_total += a;
            req_total += b;
        }
        if (push_rel) released[rel_base + rel_before + uint32_t(__popcll(rel_bits & below))] = rel_c;
        if (push_req) requested[req_base + req_before + uint32_t(__popcll(req_bits & below))] = req_c;
        rel_base += rel_total;
        req_base += req_total;
    }
    if (tid == 0) {
        counts[0] = rel_base;
        counts[1] = req_base;
    }
}

// ---- adjust_to_tile_atlas ----------------------------------------------------------------------------------------

__device__ __forceinline__ bool table_find(const StateSlot* __restrict__ table, uint32_t mask, bt_tile_coordinate c, uint32_t& atlas_index, uint32_t& loaded) {
    for (uint32_t i = hash_coordinate(c) & mask, probes = 0; probes <= mask; i = (i + 1u) & mask, probes++) {
        const StateSlot s = table[i];
        if (s.coordinate.side == kInvalid) return false;
        if (s.coordinate.side == c.side && s.coordinate.lod == c.lod && s.coordinate.x == c.x && s.coordinate.y == c.y) {
            atlas_index = s.atlas_index;
            loaded = s.loaded;
            return true;
        }
    }
    return false;
}

// TileAtlasState::get_best_tile (tile_atlas.rs:478-503) for every node
__global__ __launch_bounds__(256) void tile_tree_adjust_kernel(const NodeState* __restrict__ nodes, bt_tile_tree_entry* __restrict__ entries, uint32_t total,
                                                               const StateSlot* __restrict__ table, uint32_t mask) {
    const uint32_t n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= total) return;
    bt_tile_coordinate c = nodes[n].coordinate;
    bt_tile_tree_entry e = {BT_INVALID_ATLAS_INDEX, BT_INVALID_LOD};
    while (!(c.side == kInvalid && c.lod == kInvalid && c.x == kInvalid && c.y == kInvalid) && c.lod != BT_INVALID_LOD) {
        uint32_t atlas_index, loaded;
        if (table_find(table, mask, c, atlas_index, loaded) && loaded) {
            e = {atlas_index, c.lod};
            break;
        }
        c = {c.side, c.lod - 1u, c.x >> 1, c.y >> 1};  // parent(): wrapping_sub
    }
    entries[n] = e;
}

// ---- sampling ----------------------------------------------------------------------------------------------------

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

struct Lookup {  // TileLookup (tile_tree.rs:67-81)
    uint32_t atlas_index, atlas_lod;
    float uv[2];
};

// TileTree::lookup_tile (tile_tree.rs:241-266)
__device__ __forceinline__ Lookup lookup_tile(const TreeParams& P, const bt_tile_tree_entry* __restrict__ entries, V3 world_position, uint32_t tree_lod) {
    const Coordinate c = coordinate_from_world_position(world_position, P.model);
    const double tile_count = double(1u << tree_lod);
    const V2 t = compute_tree_xy(c, tile_count);
    const uint32_t ts = P.tree_size;
    const uint64_t ix = uint64_t(t.x), iy = uint64_t(t.y);  // `as usize` (non-negative here)
    const bt_tile_tree_entry e = entries[((c.side * P.lod_count + tree_lod) * ts + uint32_t(ix % ts)) * ts + uint32_t(iy % ts)];
    if (e.atlas_lod == BT_INVALID_LOD) return {BT_INVALID_ATLAS_INDEX, BT_INVALID_LOD, {0.0f, 0.0f}};
    const double div = double(1u << (tree_lod - e.atlas_lod));
    const double qx = t.x / div, qy = t.y / div;
    return {e.atlas_index, e.atlas_lod, {float(qx - trunc(qx)), float(qy - trunc(qy))}};  // `% 1.0`, as_vec2
}

// AtlasAttachment::sample + AttachmentData::sample (tile_atlas.rs:249-258, terrain_data/mod.rs:220-263); the same
// code as bt_kernels.hip sample_kernel
__device__ __forceinline__ void sample_lookup(const AttachmentMeta& m, const void* __restrict__ atlas, const Lookup& l, float r[4]) {
    if (l.atlas_index >= m.atlas_size) {
        r[0] = r[1] = r[2] = r[3] = 0.0f;
        return;
    }
    const uint32_t T = m.texture_size;
    const float scale = float(m.center_size) / float(T), offset = float(m.border_size) / float(T);
    float rem[2];
    int ixy[2];
#pragma unroll
    for (int a = 0; a < 2; a++) {
        const float u = l.uv[a] * scale + offset;
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
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float a = v[0][0][k] + (v[0][1][k] - v[0][0][k]) * rem[1];
        const float b = v[1][0][k] + (v[1][1][k] - v[1][0][k]) * rem[1];
        r[k] = a + (b - a) * rem[0];
    }
}

// sample_attachment / sample_height (terrain_data/mod.rs:265-307), one world position per thread
__global__ __launch_bounds__(128) void tile_tree_sample_kernel(TreeParams P, const bt_tile_tree_entry* __restrict__ entries, AttachmentMeta m,
                                                               const void* __restrict__ atlas, const double* __restrict__ positions, uint32_t count,
                                                               float4* __restrict__ out, float* heights, const float* height) {  // (heights / height may be ONE buffer — enqueue_height reads the old height, then overwrites it: no __restrict__)
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const float approximate_height = height ? *height : P.approximate_height;
    const V3 p = {positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]};
    const V3 surface = surface_position(P.model, p, double(approximate_height));
    // compute_blend (tile_tree.rs:223-239)
    const double view_distance = distance3(P.view_world_position, surface);
    const double cap = double(P.lod_count) - 0.00001;
    const double l2 = log2(P.blend_distance / view_distance);
    const float target_lod = float(l2 < cap ? l2 : cap);
    const uint32_t lod = !(target_lod > 0.0f) ? 0u : uint32_t(target_lod);  // `as u32` saturates
    float ratio = 0.0f;
    if (lod != 0) {  // inverse_mix(lod + blend_range, lod, target_lod) (util.rs:8-10)
        const float a = float(lod) + P.blend_range, b = float(lod);
        const float q = (target_lod - a) / (b - a);
        ratio = q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
    }
    float value[4];
    sample_lookup(m, atlas, lookup_tile(P, entries, surface, lod), value);
    if (ratio > 0.0f) {
        float value2[4];
        sample_lookup(m, atlas, lookup_tile(P, entries, surface, lod - 1u), value2);
#pragma unroll
        for (int k = 0; k < 4; k++) value[k] = value[k] + (value2[k] - value[k]) * ratio;  // Vec4::lerp
    }
    out[i] = make_float4(value[0], value[1], value[2], value[3]);
    if (heights) heights[i] = P.model.min_height + (P.model.max_height - P.model.min_height) * value[0];  // f32::lerp
}

}  // namespace

struct bt_tile_tree {
    bt_ctx* ctx = nullptr;
    bt_terrain_model model_c{};
    bt_terrain_view_config view_config{};
    Model model{};
    uint32_t lod_count = 0, sides = 0, nodes = 0;
    // TileTree::new (tile_tree.rs:135-173)
    double morph_distance = 0, blend_distance = 0, load_distance = 0, subdivision_distance = 0, precision_threshold_distance = 0;
    double view_world_position[3] = {0, 0, 0};
    float approximate_height = 0;
    // device tables
    NodeState* d_nodes = nullptr;
    bt_tile_tree_entry* d_entries = nullptr;
    uint32_t* d_origins = nullptr;
    float* d_height = nullptr;  // [0] approximate_height (what the kernels read), [4..8) the vec4 sample behind it
    StateSlot* d_table = nullptr;
    uint32_t table_capacity = 0;
    uint64_t table_version = 0;
    const bt_atlas* table_atlas = nullptr;
    // the last update's lists: pinned host memory the update kernel writes directly (one synchronisation, no copies)
    bt_tile_coordinate *h_released = nullptr, *h_requested = nullptr;
    uint32_t* h_counts = nullptr;
    uint32_t released_count = 0, requested_count = 0;
    // bt_frame_update: pinned staging of the view position, of the height read back one frame late, and of the atlas's tile states
    double* h_view_position = nullptr;
    float* h_height = nullptr;
    bool height_in_flight = false;  // an asynchronous copy d_height -> h_height was enqueued after the last synchronisation
    StateSlot* h_table = nullptr;
    uint32_t h_table_capacity = 0;
    bool table_copy_pending = false;  // h_table -> d_table enqueued since the last synchronisation
};

namespace {

TreeParams make_params(const bt_tile_tree* t) {
    TreeParams P{};
    P.model = t->model;
    P.lod_count = t->lod_count;
    P.tree_size = t->view_config.tree_size;
    P.sides = t->sides;
    P.load_distance = t->load_distance;
    P.blend_distance = t->blend_distance;
    P.blend_range = t->view_config.blend_range;
    P.approximate_height = t->approximate_height;
    P.view_world_position = {t->view_world_position[0], t->view_world_position[1], t->view_world_position[2]};
    const Coordinate vc = coordinate_from_world_position(P.view_world_position, t->model);
    for (uint32_t s = 0; s < 6; s++) P.view_coordinate[s] = s < t->sides ? coordinate_project_to_side(vc, s, t->model) : Coordinate{s, {0.0, 0.0}};
    return P;
}

bt_status check_model(const bt_terrain_model* m) {
    if (!m || m->kind > BT_MODEL_ELLIPSOIDAL || !(m->a > 0.0) || (m->kind == BT_MODEL_ELLIPSOIDAL && !(m->b > 0.0))) {
        set_error("terrain model: kind / axes");
        return BT_ERR_INVALID_ARGUMENT;
    }
    return BT_OK;
}

}  // namespace

extern "C" {

void bt_terrain_view_config_default(bt_terrain_view_config* out) {
    if (!out) return;
    *out = {8, 1000000, 30, 16, 0.1, 0.001, 2.5, 16.0, 2.0, 0.2f, 0.2f, 10, 0};  // terrain_view.rs:47-63
}

bt_status bt_view_state_from_config(const bt_terrain_model* model, const bt_terrain_view_config* vc, const double view_world_position[3],
                                    float approximate_height, bt_view_state* out) {
    if (!vc || !view_world_position || !out) return BT_ERR_INVALID_ARGUMENT;
    if (bt_status s = check_model(model)) return s;
    const Model m = make_model(*model);
    bt_view_state v{};
    v.spherical = is_spherical(m) ? 1u : 0u;
    v.geometry_tile_count = vc->geometry_tile_count;
    v.refinement_count = vc->refinement_count;
    v.vertices_per_tile = 2u * vc->grid_size * (vc->grid_size + 2u);  // terrain_view_bind_group.rs:106
    // tile_tree.rs:148-150 (f64), `as f32` at terrain_view_bind_group.rs:111
    v.subdivision_distance = float(vc->morph_distance * model_scale(m) * (1.0 + vc->subdivision_tolerance));
    v.origin_lod = vc->origin_lod;
    v.approximate_height = approximate_height;
    // TerrainModelApproximation::compute (terrain_model.rs:262-290): origin_xy / origin_uv per side
    const V3 view = {view_world_position[0], view_world_position[1], view_world_position[2]};
    const Coordinate c = coordinate_from_world_position(view, m);
    const double origin_count = double(1u << vc->origin_lod);
    for (uint32_t side = 0; side < 6; side++) {
        // (the planar model fills all six entries with the same coordinate: project_to_side returns self)
        const Coordinate p = coordinate_project_to_side(c, side, m);
        const double sx = p.uv.x * origin_count, sy = p.uv.y * origin_count;
        v.sides[side].view_xy[0] = saturating_i32(sx);  // as_ivec2
        v.sides[side].view_xy[1] = saturating_i32(sy);
        v.sides[side].view_uv[0] = float(sx - trunc(sx));  // DVec2::fract() = self - self.trunc() (glam 0.27), as_vec2
        v.sides[side].view_uv[1] = float(sy - trunc(sy));
    }
    for (int i = 0; i < 3; i++) v.world_position[i] = float(view_world_position[i]);  // culling_bind_group.rs:50
    // mesh uniform of TerrainModel::transform() (terrain_model.rs:195-201): scale / translation `as_vec3`, identity
    // rotation; local_from_world_transpose = transpose(inverse(mat3)) = diag(1 / scale) in f32
    const float sc[3] = {float(m.scale.x), float(m.scale.y), float(m.scale.z)};
    const float tr[3] = {float(m.position.x), float(m.position.y), float(m.position.z)};
    for (int col = 0; col < 3; col++) v.world_from_local[3 * col + col] = sc[col];
    for (int i = 0; i < 3; i++) v.world_from_local[9 + i] = tr[i];
    for (int col = 0; col < 3; col++) v.local_from_world_transpose[3 * col + col] = 1.0f / sc[col];
    *out = v;
    return BT_OK;
}

bt_status bt_tile_tree_create(bt_ctx* ctx, const bt_terrain_model* model, uint32_t lod_count, const bt_terrain_view_config* vc, bt_tile_tree** out) {
    if (!ctx || !vc || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (bt_status s = check_model(model)) return s;
    if (lod_count == 0 || lod_count > 31 || vc->tree_size == 0 || vc->tree_size > 64) {
        set_error("tile tree: lod_count %u / tree_size %u", lod_count, vc->tree_size);
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(ctx->device));
    bt_tile_tree* t = new bt_tile_tree();
    t->ctx = ctx;
    t->model_c = *model;
    t->view_config = *vc;
    t->model = make_model(*model);
    t->lod_count = lod_count;
    t->sides = side_count(t->model);
    t->nodes = t->sides * lod_count * vc->tree_size * vc->tree_size;
    const double scale = model_scale(t->model);
    t->morph_distance = vc->morph_distance * scale;
    t->blend_distance = vc->blend_distance * scale;
    t->load_distance = vc->load_distance * scale;
    t->subdivision_distance = vc->morph_distance * scale * (1.0 + vc->subdivision_tolerance);
    t->precision_threshold_distance = vc->precision_threshold_distance * scale;
    t->approximate_height = (model->min_height + model->max_height) / 2.0f;
    hipError_t e = hipMalloc((void**)&t->d_nodes, sizeof(NodeState) * t->nodes);
    if (e == hipSuccess) e = hipMalloc((void**)&t->d_entries, sizeof(bt_tile_tree_entry) * t->nodes);
    if (e == hipSuccess) e = hipMalloc((void**)&t->d_origins, sizeof(uint32_t) * 2 * t->sides * lod_count);
    if (e == hipSuccess) e = hipMalloc((void**)&t->d_height, sizeof(float) * 8);
    if (e == hipSuccess) e = hipHostMalloc((void**)&t->h_view_position, sizeof(double) * 3, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&t->h_height, sizeof(float), hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&t->h_released, sizeof(bt_tile_coordinate) * t->nodes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&t->h_requested, sizeof(bt_tile_coordinate) * t->nodes, hipHostMallocDefault);
    if (e == hipSuccess) e = hipHostMalloc((void**)&t->h_counts, sizeof(uint32_t) * 2, hipHostMallocDefault);
    // TileState::default (coordinate INVALID, Released), TileTreeEntry::default (INVALID, INVALID), origins 0
    if (e == hipSuccess) e = hipMemsetAsync(t->d_nodes, 0xFF, sizeof(NodeState) * t->nodes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(t->d_entries, 0xFF, sizeof(bt_tile_tree_entry) * t->nodes, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(t->d_origins, 0, sizeof(uint32_t) * 2 * t->sides * lod_count, ctx->stream);
    if (e != hipSuccess) {
        bt_tile_tree_destroy(t);
        return hip_fail(e, "tile tree allocation");
    }
    // `requested` must start as 0 (Released): the 0xFF fill set it; clear that field with one small kernel-free pass
    std::vector<NodeState> init(t->nodes, NodeState{{kInvalid, kInvalid, kInvalid, kInvalid}, 0u});
    e = hipMemcpyAsync(t->d_nodes, init.data(), sizeof(NodeState) * t->nodes, hipMemcpyHostToDevice, ctx->stream);
    const float height0[8] = {t->approximate_height, 0, 0, 0, 0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(t->d_height, height0, sizeof height0, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) {
        bt_tile_tree_destroy(t);
        return hip_fail(e, "tile tree initialisation");
    }
    *out = t;
    return BT_OK;
}

void bt_tile_tree_destroy(bt_tile_tree* t) {
    if (!t) return;
    hipSetDevice(t->ctx->device);
    for (void* p : {(void*)t->d_nodes, (void*)t->d_entries, (void*)t->d_origins, (void*)t->d_height, (void*)t->d_table})
        if (p) hipFree(p);
    for (void* p : {(void*)t->h_released, (void*)t->h_requested, (void*)t->h_counts, (void*)t->h_view_position, (void*)t->h_height, (void*)t->h_table})
        if (p) hipHostFree(p);
    delete t;
}

namespace {
// the host's mirror of the tree's height: refreshed whenever the stream has just been synchronised
void adopt_height(bt_tile_tree* t) {
    if (t->height_in_flight) {
        t->approximate_height = *t->h_height;
        t->height_in_flight = false;
    }
}

// TileTree::update as one launch; the kernel writes both lists and their lengths straight into pinned host memory
bt_status enqueue_update(bt_tile_tree* t, const double view_world_position[3]) {
    for (int i = 0; i < 3; i++) t->view_world_position[i] = view_world_position[i];
    const TreeParams P = make_params(t);
    tile_tree_update_kernel<<<1, kThreads, 0, t->ctx->stream>>>(P, t->d_nodes, t->d_origins, t->h_released, t->h_requested, t->h_counts, t->d_height);
    BT_HIP(hipGetLastError());
    return BT_OK;
}

bt_status finish_update(bt_tile_tree* t) {
    BT_HIP(hipStreamSynchronize(t->ctx->stream));
    adopt_height(t);
    t->table_copy_pending = false;
    t->released_count = t->h_counts[0];
    t->requested_count = t->h_counts[1];
    return BT_OK;
}
}  // namespace

bt_status bt_tile_tree_update(bt_tile_tree* t, const double view_world_position[3]) {
    if (!t || !view_world_position) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(t->ctx->device));
    if (bt_status s = enqueue_update(t, view_world_position)) return s;
    return finish_update(t);
}

bt_status bt_tile_tree_requests(const bt_tile_tree* t, const bt_tile_coordinate** released, uint32_t* released_count, const bt_tile_coordinate** requested,
                                uint32_t* requested_count) {
    if (!t) return BT_ERR_INVALID_ARGUMENT;
    if (released) *released = t->h_released;
    if (released_count) *released_count = t->released_count;
    if (requested) *requested = t->h_requested;
    if (requested_count) *requested_count = t->requested_count;
    return BT_OK;
}

bt_status bt_tile_tree_apply_requests(bt_tile_tree* t, bt_atlas* a) {
    if (!t || !a) return BT_ERR_INVALID_ARGUMENT;
    // tile_atlas.rs:590-600: all releases of the tree, then all its requests.  The reference DRAINS both lists whatever
    // happens (mem::take); here too: an entry is consumed when it is applied and, on an error (e.g. "Atlas out of indices"),
    // the rest of both lists is dropped — a retry can neither release twice nor count a request twice.  The failing entry's
    // status is returned; tiles not requested because of it are requested again by a later update() if still wanted.
    bt_status first = BT_OK;
    for (uint32_t i = 0; i < t->released_count && first == BT_OK; i++) first = bt_atlas_release_tile(a, t->h_released[i]);
    for (uint32_t i = 0; i < t->requested_count && first == BT_OK; i++) first = bt_atlas_request_tile(a, t->h_requested[i]);
    t->released_count = 0;
    t->requested_count = 0;
    return first;
}

bt_status bt_tile_tree_adjust_to_tile_atlas(bt_tile_tree* t, const bt_atlas* a) {
    if (!t || !a) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    if (t->table_atlas != a || t->table_version != a->state_version || !t->d_table) {
        uint32_t capacity = 64;
        while (capacity < 2 * a->tile_states.size() + 2) capacity *= 2;
        if (capacity > t->h_table_capacity) {
            // (the staging table is rewritten below: a copy out of the old one may still be in flight)
            BT_HIP(hipStreamSynchronize(s));
            adopt_height(t);
            if (t->h_table) BT_HIP(hipHostFree(t->h_table));
            t->h_table = nullptr;
            t->h_table_capacity = 0;
            BT_HIP(hipHostMalloc((void**)&t->h_table, sizeof(StateSlot) * capacity, hipHostMallocDefault));
            t->h_table_capacity = capacity;
        }
        // The pinned staging table is reused by every upload.  Its previous copy was enqueued before the update kernel whose
        // lists the host has since read (bt_tile_tree_update / bt_frame_update synchronise on them), i.e. it has completed —
        // unless the caller adjusts twice without an update in between: then wait for it.
        if (t->table_copy_pending) BT_HIP(hipStreamSynchronize(s));
        StateSlot* table = t->h_table;
        for (uint32_t i = 0; i < capacity; i++) table[i] = StateSlot{{kInvalid, kInvalid, kInvalid, kInvalid}, BT_INVALID_ATLAS_INDEX, 0u};
        for (const auto& kv : a->tile_states) {
            uint32_t i = hash_coordinate(kv.first) & (capacity - 1);
            while (table[i].coordinate.side != kInvalid) i = (i + 1u) & (capacity - 1);
            table[i] = {kv.first, kv.second.atlas_index, kv.second.loading == 0 ? 1u : 0u};
        }
        if (capacity > t->table_capacity) {
            if (t->d_table) BT_HIP(hipFree(t->d_table));
            t->d_table = nullptr;
            BT_HIP(hipMalloc((void**)&t->d_table, sizeof(StateSlot) * capacity));
        }
        t->table_capacity = capacity;
        BT_HIP(hipMemcpyAsync(t->d_table, table, sizeof(StateSlot) * capacity, hipMemcpyHostToDevice, s));  // pinned source: asynchronous
        t->table_copy_pending = true;
        t->table_atlas = a;
        t->table_version = a->state_version;
    }
    tile_tree_adjust_kernel<<<(t->nodes + 255u) / 256u, 256, 0, s>>>(t->d_nodes, t->d_entries, t->nodes, t->d_table, t->table_capacity - 1u);
    BT_HIP(hipGetLastError());
    return BT_OK;
}

bt_status bt_tile_tree_buffers(const bt_tile_tree* t, void** entries, void** origins) {
    if (!t) return BT_ERR_INVALID_ARGUMENT;
    if (entries) *entries = t->d_entries;
    if (origins) *origins = t->d_origins;
    return BT_OK;
}

bt_status bt_tile_tree_read(bt_tile_tree* t, bt_tile_tree_entry* entries, uint32_t entry_cap, uint32_t* origins_xy, uint32_t origin_cap,
                            bt_tile_coordinate* node_coordinates, uint32_t* node_requested) {
    if (!t) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    if (entries) BT_HIP(hipMemcpyAsync(entries, t->d_entries, sizeof(bt_tile_tree_entry) * std::min(entry_cap, t->nodes), hipMemcpyDeviceToHost, s));
    if (origins_xy) BT_HIP(hipMemcpyAsync(origins_xy, t->d_origins, sizeof(uint32_t) * std::min(origin_cap, 2 * t->sides * t->lod_count), hipMemcpyDeviceToHost, s));
    std::vector<NodeState> nodes;
    if (node_coordinates || node_requested) {
        nodes.resize(t->nodes);
        BT_HIP(hipMemcpyAsync(nodes.data(), t->d_nodes, sizeof(NodeState) * t->nodes, hipMemcpyDeviceToHost, s));
    }
    BT_HIP(hipStreamSynchronize(s));
    adopt_height(t);
    t->table_copy_pending = false;
    for (uint32_t i = 0; i < uint32_t(nodes.size()) && i < entry_cap; i++) {
        if (node_coordinates) node_coordinates[i] = nodes[i].coordinate;
        if (node_requested) node_requested[i] = nodes[i].requested;
    }
    return BT_OK;
}

bt_status bt_tile_tree_sample_attachment(bt_tile_tree* t, bt_atlas* a, uint32_t ai, const double* positions, uint32_t count, float* out_vec4, float* heights) {
    if (!t || !a || ai >= a->attachments.size() || (count && (!positions || !out_vec4))) return BT_ERR_INVALID_ARGUMENT;
    const Attachment& at = a->attachments[ai];
    if (at.meta.format != BT_FORMAT_R16 && at.meta.format != BT_FORMAT_RGBA8) return BT_ERR_UNSUPPORTED;
    if (!count) return BT_OK;
    BT_HIP(hipSetDevice(t->ctx->device));
    hipStream_t s = t->ctx->stream;
    uint8_t* dev = nullptr;
    const size_t in_bytes = sizeof(double) * 3 * size_t(count), out_bytes = 16 * size_t(count), h_bytes = 4 * size_t(count);
    BT_HIP(hipMalloc((void**)&dev, in_bytes + out_bytes + h_bytes));
    hipError_t e = hipMemcpyAsync(dev, positions, in_bytes, hipMemcpyHostToDevice, s);
    if (e == hipSuccess) {
        tile_tree_sample_kernel<<<(count + 127u) / 128u, 128, 0, s>>>(make_params(t), t->d_entries, at.meta, at.level0, (const double*)dev, count,
                                                                      (float4*)(dev + in_bytes), (float*)(dev + in_bytes + out_bytes), t->d_height);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out_vec4, dev + in_bytes, out_bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && heights) e = hipMemcpyAsync(heights, dev + in_bytes + out_bytes, h_bytes, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "bt_tile_tree_sample_attachment");
    return BT_OK;
}

namespace {
// TileTree::approximate_height (tile_tree.rs:372-386): sample_height of attachment 0 at the view position, kept on the device
// (d_height[0]; the kernel reads the old value for its surface position, then overwrites it) and copied to pinned memory
bt_status enqueue_height(bt_tile_tree* t, bt_atlas* a) {
    if (a->attachments.empty()) return BT_ERR_INVALID_ARGUMENT;
    const Attachment& at = a->attachments[0];
    if (at.meta.format != BT_FORMAT_R16 && at.meta.format != BT_FORMAT_RGBA8) return BT_ERR_UNSUPPORTED;
    for (int i = 0; i < 3; i++) t->h_view_position[i] = t->view_world_position[i];  // (read by the kernel: the previous kernel that read it has finished — the caller synchronised on the update since)
    hipStream_t s = t->ctx->stream;
    tile_tree_sample_kernel<<<1, 128, 0, s>>>(make_params(t), t->d_entries, at.meta, at.level0, t->h_view_position, 1u, (float4*)(t->d_height + 4), t->d_height, t->d_height);
    BT_HIP(hipGetLastError());
    BT_HIP(hipMemcpyAsync(t->h_height, t->d_height, sizeof(float), hipMemcpyDeviceToHost, s));
    t->height_in_flight = true;
    return BT_OK;
}
}  // namespace

bt_status bt_tile_tree_approximate_height(bt_tile_tree* t, bt_atlas* a, float* height) {
    if (!t || !a) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(t->ctx->device));
    if (bt_status s = enqueue_height(t, a)) return s;
    BT_HIP(hipStreamSynchronize(t->ctx->stream));
    adopt_height(t);
    t->table_copy_pending = false;
    if (height) *height = t->approximate_height;
    return BT_OK;
}

// The reference's per-frame chain (src/plugin.rs:46-56): TileTree::compute_requests -> TileAtlas::update (the tree's releases
// and requests) -> TileTree::adjust_to_tile_atlas -> TileTree::approximate_height -> the tiling prepass of the view, as ONE call
// with ONE host synchronisation: the one the reference's structure forces, because the atlas's streaming state machine (LRU,
// file loads) is host code and needs the two lists.  Everything behind it is enqueued and left running: the tile states travel
// from pinned memory, the height stays on the device (the prepass kernels and the next frame's update read it there; the host's
// copy arrives with the next synchronisation), the final tiles and indirect arguments are in the buffers a renderer binds.
bt_status bt_frame_update(bt_tile_tree* t, bt_atlas* a, bt_tiling_prepass* prepass, const double view_world_position[3], uint32_t flags, bt_frame_info* out) {
    if (!t || !a || !view_world_position) return BT_ERR_INVALID_ARGUMENT;
    if (t->ctx != a->ctx) {
        set_error("tile tree and atlas belong to different contexts");
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (prepass && bt::tiling_prepass_ctx(prepass) != t->ctx) {
        // the prepass kernels read the height the sample kernel of THIS call leaves on the device: ordered only on one stream
        set_error("tile tree and tiling prepass belong to different contexts");
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(t->ctx->device));
    if (bt_status s = enqueue_update(t, view_world_position)) return s;
    if (bt_status s = finish_update(t)) return s;  // the synchronisation
    bt_frame_info info{};
    info.released_count = t->released_count;
    info.requested_count = t->requested_count;
    info.approximate_height = t->approximate_height;  // what this frame's update saw
    if (!(flags & BT_FRAME_KEEP_REQUESTS)) info.apply_status = bt_tile_tree_apply_requests(t, a);
    if (bt_status s = bt_tile_tree_adjust_to_tile_atlas(t, a)) return s;
    if (!(flags & BT_FRAME_KEEP_HEIGHT))
        if (bt_status s = enqueue_height(t, a)) return s;
    if (prepass) {
        bt_view_state view;
        if (bt_status s = bt_tile_tree_view_state(t, &view)) return s;  // (approximate_height: the host's copy; the kernels take the device's)
        const uint32_t form = (flags & BT_FRAME_PREPASS_UNORDERED) ? 1u : ((flags & BT_FRAME_PREPASS_PLAIN) ? 2u : 0u);
        if (bt_status s = bt::tiling_prepass_enqueue(prepass, &view, t->d_height, form)) return s;
    }
    if (out) *out = info;
    return BT_OK;
}

bt_status bt_tile_tree_view_state(const bt_tile_tree* t, bt_view_state* out) {
    if (!t || !out) return BT_ERR_INVALID_ARGUMENT;
    return bt_view_state_from_config(&t->model_c, &t->view_config, t->view_world_position, t->approximate_height, out);
}

}  // extern "C"
