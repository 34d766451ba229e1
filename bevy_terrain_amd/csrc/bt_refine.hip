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
    uint32_t* counters = nullptr;  // [0] final count, [1] overflow flag, [2] tiles visited, [3] passes; unordered form: [8..40) tiles visited per LOD, [40..72) dividing tiles per LOD
    unsigned long long* bits = nullptr;  // divide bits of every (side, lod) window (allocated on first use)
    int window = 0;                      // window radius of the unordered form (0 = the default, kWinK)
    bool unordered = false;              // the last run was the unordered form: read() derives counters [1..3] from the per-LOD counts
    uint32_t unordered_capacity = 0;
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
// approximate_height: the view's (bt_view_state::approximate_height), or the value bt_frame_update left on the device — passed
// beside the view: writing it into the by-value kernel argument makes the compiler copy the whole struct to scratch (224 bytes,
// 14 -> 52 VGPRs in the divide-bits kernel: measured, round 4)
__device__ bool should_be_divided(const bt_view_state& v, const bt_tile_coordinate& tile, float approximate_height) {
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
    const float dx = (wx + approximate_height * nx) - v.world_position[0];
    const float dy = (wy + approximate_height * ny) - v.world_position[1];
    const float dz = (wz + approximate_height * nz) - v.world_position[2];
    const float view_distance = length3(dx, dy, dz);
    return view_distance < v.subdivision_distance * inv_tc;
}

// ---- the divide test of every tile that can matter, computed up front ------------------------------------------------
// The schedule is a chain of up to refinement_count + 1 dependent passes, and a pass of the single-workgroup kernel costs
// ~2.5 us of pure latency: tile load (L2) -> ~200 dependent VALU instructions of should_be_divided -> ballot / scan /
// barrier.  But should_be_divided(tile) depends on the tile and the view alone, not on the pass: so a first launch
// evaluates it SPECULATIVELY for every tile inside a (2K+1)^2 window around the view's tile at every LOD of every side
// (tiles only divide close to the view: the criterion scales with the tile size) — ~1.3 k workgroups, one evaluation per
// thread, all independent — and leaves one bit per tile.  The ordered kernel then reads bits (from LDS) instead of running
// the arithmetic, keeps its frontier in LDS while it fits, and a pass shrinks to a few hundred nanoseconds.  A tile outside
// its window (possible, rare) is evaluated in place with the same function: the result is the same list in the same order.
constexpr int kWinK = 28;  // window radius in tiles: tiles DIVIDE within ~10 tiles of the view, so tiles EXIST within ~2 x 10 + 2 (more under the cube-sphere warp)
constexpr uint32_t kWinW = 2 * kWinK + 1, kWinBits = kWinW * kWinW;
constexpr uint32_t kWinChunks = (kWinBits + 255) / 256;      // 256-thread workgroups per (side, lod)
constexpr uint32_t kWinWords = kWinChunks * 4;               // 64-bit words per (side, lod): one per wave
constexpr uint32_t kMaxLods = 32;
constexpr uint32_t kFrontierCap = 2048;                      // tiles of a pass kept in LDS

// first tile (x, y) of the window of (side, lod): centred on the view's tile of that LOD, clamped into the face
__device__ __forceinline__ void window_origin(const bt_view_state& v, uint32_t side, uint32_t lod, int& ox, int& oy, int radius = kWinK) {
    // view_xy is signed: on a neighbouring cube face the view's tile lies outside [0, 2^origin_lod); the window belongs at the
    // face's NEAR edge then (cast to u32 first, a negative coordinate would end up clamped to the far one)
    const int olast = int((1u << min(v.origin_lod, 30u)) - 1u);
    const int vx = min(max(v.sides[side].view_xy[0], 0), olast), vy = min(max(v.sides[side].view_xy[1], 0), olast);
    Coordinate vc{side, v.origin_lod, uint32_t(vx), uint32_t(vy), v.sides[side].view_uv[0], v.sides[side].view_uv[1]};
    coordinate_change_lod(vc, lod);
    const int last = int((1u << lod) - 1u);  // lod <= 31
    const int cx = min(max(int(vc.x), 0), last), cy = min(max(int(vc.y), 0), last);
    ox = cx - radius;
    oy = cy - radius;
}

// lods: LODs 0 .. lods - 1 get their bits (the host's estimate of how deep this view can refine; anything deeper is
// evaluated in place by the ordered kernel — an estimate can only cost time, never change the result)
constexpr uint32_t kCntVisited = 8, kCntDivide = 40, kCounterWords = 72;  // (bt_tiling_prepass::counters)

// radius <= kWinK: the window actually used (storage is laid out for kWinK); counters != nullptr: also resets the counters and
// the indirect arguments the unordered collector (next launch) accumulates into
__global__ __launch_bounds__(256) void tiling_divide_bits_kernel(bt_view_state view, uint32_t lods, int radius, unsigned long long* __restrict__ bits,
                                                                 uint32_t* __restrict__ counters, bt_indirect* __restrict__ indirect, const float* __restrict__ height) {
    const float approximate_height = height ? *height : view.approximate_height;  // (bt_frame_update: the height sampled earlier on this stream, never seen by the host)
    const uint32_t W = 2u * uint32_t(radius) + 1u, chunks = (W * W + 255u) / 256u;
    const uint32_t chunk = blockIdx.x % chunks, lod = (blockIdx.x / chunks) % lods, side = blockIdx.x / (chunks * lods);
    if (counters && blockIdx.x == 0) {
        if (threadIdx.x < kCounterWords) counters[threadIdx.x] = 0;
        if (threadIdx.x == 0) *indirect = {0u, 1u, 0u, 0u};  // prepare_render's instance_count / bases; the vertex count accumulates
    }
    if (lod > view.refinement_count) return;
    int ox, oy;
    window_origin(view, side, lod, ox, oy, radius);
    const uint32_t b = chunk * 256u + threadIdx.x;
    const int tx = ox + int(b % W), ty = oy + int(b / W), last = int((1u << lod) - 1u);
    bool divide = false;
    if (b < W * W && tx >= 0 && ty >= 0 && tx <= last && ty <= last) divide = should_be_divided(view, bt_tile_coordinate{side, lod, uint32_t(tx), uint32_t(ty)}, approximate_height);
    const unsigned long long word = __ballot(divide);
    if ((threadIdx.x & 63u) == 0) bits[(size_t(side) * kMaxLods + lod) * kWinWords + chunk * 4u + (threadIdx.x >> 6)] = word;
}

// ---- the unordered form: the final SET straight from the bits ---------------------------------------------------------
// The reference appends final tiles in the arrival order of global atomics (refine_tiles.wgsl:13-15, 41): its contract is the
// set, not the order.  With every divide test known, membership needs no passes at all: a tile is visited iff all its
// ancestors divide, and it is final iff it is visited, does not divide itself and its LOD is a pass that runs
// (lod <= refinement_count; the children of tiles that still divide in the last pass are dropped, as in the reference).
// One thread per window tile walks its <= 31 ancestors' bits in LDS; finals are appended with one atomic per wave.  A
// visited, dividing tile whose child lies outside the child LOD's window (or deeper than the LODs that got bits) owns that
// child's whole subtree: a stackless depth-first walk evaluating should_be_divided in place, skipping anything that is
// inside a window (those tiles have their own thread) — so every tile is decided exactly once, whatever the window size.
// prepare_render's vertex count (prepare_prepass.wgsl:38-44) accumulates with the same reservations; per-LOD visit / divide
// counts let bt_tiling_prepass_read give the overflow verdict of the reference's buffers (a pass needs parents + children
// <= N, prepare_prepass.wgsl:25-36; the final list <= N).  No ticket, no fence: a "last workgroup" protocol made ~1500
// same-address atomics the longest thing in the kernel.

struct WindowBits {
    const unsigned long long* bits;  // LDS: [lod][kWinWords] of one side
    const int2* origin;              // LDS: [lod]
    uint32_t lods;
    int W;
    __device__ __forceinline__ bool inside(uint32_t lod, uint32_t x, uint32_t y, uint32_t& b) const {
        if (lod >= lods) return false;
        const int bx = int(x) - origin[lod].x, by = int(y) - origin[lod].y;
        b = uint32_t(by * W + bx);
        return bx >= 0 && by >= 0 && bx < W && by < W;
    }
    __device__ __forceinline__ bool bit(uint32_t lod, uint32_t b) const { return (bits[lod * kWinWords + (b >> 6)] >> (b & 63u)) & 1ull; }
};

__global__ __launch_bounds__(256) void tiling_collect_kernel(bt_view_state view, uint32_t lods, int radius, uint32_t capacity,
                                                             const unsigned long long* __restrict__ bits, bt_tile_coordinate* __restrict__ final_tiles,
                                                             bt_indirect* __restrict__ indirect, uint32_t* __restrict__ counters, const float* __restrict__ height) {
    __shared__ unsigned long long s_bits[kMaxLods * kWinWords];
    const float approximate_height = height ? *height : view.approximate_height;
    __shared__ int2 s_origin[kMaxLods];
    const uint32_t W = 2u * uint32_t(radius) + 1u, chunks = (W * W + 255u) / 256u;
    const uint32_t chunk = blockIdx.x % chunks, lod = (blockIdx.x / chunks) % lods, side = blockIdx.x / (chunks * lods);
    const uint32_t tid = threadIdx.x, lane = tid & 63u, rc = view.refinement_count;
    if (lod > rc) return;
    {
        if (tid < lods) {
            int ox, oy;
            window_origin(view, side, tid, ox, oy, radius);
            s_origin[tid] = int2{ox, oy};
        }
        const uint32_t words = min(lod + 2u, lods) * kWinWords;  // ancestors, the tile itself, its children's window
        for (uint32_t i = tid; i < words; i += 256u) s_bits[i] = bits[size_t(side) * kMaxLods * kWinWords + i];
    }
    __syncthreads();
    {
        const WindowBits wb{s_bits, s_origin, lods, int(W)};
        const uint32_t b = chunk * 256u + tid;
        const int tx = s_origin[lod].x + int(b % W), ty = s_origin[lod].y + int(b / W), last = int((1u << lod) - 1u);
        bool reach = b < W * W && tx >= 0 && ty >= 0 && tx <= last && ty <= last;
        for (uint32_t a = lod; reach && a-- > 0;) {  // all ancestors divide (order is irrelevant; an ancestor outside its window is evaluated in place)
            const uint32_t ax = uint32_t(tx) >> (lod - a), ay = uint32_t(ty) >> (lod - a);
            uint32_t ab;
            reach = wb.inside(a, ax, ay, ab) ? wb.bit(a, ab) : should_be_divided(view, bt_tile_coordinate{side, a, ax, ay}, approximate_height);
        }
        const bool divide = reach && wb.bit(lod, b);
        const bool fin = reach && !divide;
        // finals: one reservation per wave
        const unsigned long long ballot_f = __ballot(fin), ballot_r = __ballot(reach), ballot_d = __ballot(divide);
        uint32_t base = 0;
        if (lane == 0) {
            if (ballot_f) {
                base = atomicAdd(&counters[0], uint32_t(__popcll(ballot_f)));
                atomicAdd(&indirect->vertex_count, view.vertices_per_tile * uint32_t(__popcll(ballot_f)));
            }
            if (ballot_r) atomicAdd(&counters[kCntVisited + lod], uint32_t(__popcll(ballot_r)));
            if (ballot_d) atomicAdd(&counters[kCntDivide + lod], uint32_t(__popcll(ballot_d)));
        }
        base = __shfl(base, 0);
        if (fin) {
            const uint32_t fi = base + uint32_t(__popcll(ballot_f & ((1ull << lane) - 1ull)));
            if (fi < capacity) final_tiles[fi] = bt_tile_coordinate{side, lod, uint32_t(tx), uint32_t(ty)};
        }
        // children no thread owns: their subtrees, depth first, without a stack (child order 0..3 = (x & 1) | (y & 1) << 1)
        if (divide && lod < rc) {
            for (uint32_t i = 0; i < 4; i++) {
                uint32_t nl = lod + 1u, nx = (uint32_t(tx) << 1) + (i & 1u), ny = (uint32_t(ty) << 1) + (i >> 1), nb;
                if (wb.inside(nl, nx, ny, nb)) continue;
                for (;;) {
                    bool descend = false;
                    if (!wb.inside(nl, nx, ny, nb)) {
                        atomicAdd(&counters[kCntVisited + nl], 1u);
                        const bt_tile_coordinate node{side, nl, nx, ny};
                        if (should_be_divided(view, node, approximate_height)) {
                            atomicAdd(&counters[kCntDivide + nl], 1u);
                            descend = nl < rc;
                        } else {
                            const uint32_t fi = atomicAdd(&counters[0], 1u);
                            atomicAdd(&indirect->vertex_count, view.vertices_per_tile);
                            if (fi < capacity) final_tiles[fi] = node;
                        }
                    }
                    if (descend) {
                        nl++;
                        nx <<= 1;
                        ny <<= 1;
                        continue;
                    }
                    bool done = false;
                    for (;;) {  // next sibling, or up until there is one; back at the start level: finished
                        if (nl == lod + 1u) {
                            done = true;
                            break;
                        }
                        const uint32_t c = (nx & 1u) | ((ny & 1u) << 1);
                        if (c < 3u) {
                            nx = (nx & ~1u) | ((c + 1u) & 1u);
                            ny = (ny & ~1u) | ((c + 1u) >> 1);
                            break;
                        }
                        nl--;
                        nx >>= 1;
                        ny >>= 1;
                    }
                    if (done) break;
                }
            }
        }
    }
}

// kAssist = false: the whole schedule from the arithmetic alone (the plain kernel: checker, and fallback for odd views).
// kAssist = true: `bits` holds tiling_divide_bits_kernel's answers; dynamic LDS = bits of every (side, lod) + the origin
// table + two frontier buffers.
// __syncthreads() also drains the wave's global stores (one memory counter): with the frontier in LDS nothing a pass stores to
// global memory is read back inside the kernel, so its barriers only have to order LDS — the store latency (~1.5 us per
// barrier) then stays off the chain of passes
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

template <bool kAssist>
__global__ __launch_bounds__(kThreads) void tiling_prepass_kernel(bt_view_state view, uint32_t capacity,
                                                                  bt_tile_coordinate* __restrict__ temporary_tiles,
                                                                  bt_tile_coordinate* __restrict__ final_tiles,
                                                                  bt_indirect* __restrict__ indirect,
                                                                  uint32_t* __restrict__ counters, const unsigned long long* __restrict__ bits, uint32_t assist_lods,
                                                                  const float* __restrict__ height) {
    const float approximate_height = height ? *height : view.approximate_height;
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
    // kAssist: the answers of every window, the window origins and the LDS frontier
    extern __shared__ __attribute__((aligned(16))) unsigned char s_dyn[];
    const uint32_t sides = view.spherical ? 6u : 1u;
    unsigned long long* s_bits = reinterpret_cast<unsigned long long*>(s_dyn);
    int2* s_origin = reinterpret_cast<int2*>(s_bits + size_t(sides) * kMaxLods * kWinWords);
    bt_tile_coordinate* s_tiles = reinterpret_cast<bt_tile_coordinate*>(s_origin + sides * kMaxLods);  // [2][kFrontierCap]
    if constexpr (kAssist) {
        for (uint32_t i = tid; i < sides * kMaxLods * kWinWords; i += kThreads)
            if ((i / kWinWords) % kMaxLods < assist_lods) s_bits[i] = bits[i];
        if (tid < sides * kMaxLods) {
            int ox, oy;
            window_origin(view, tid / kMaxLods, tid % kMaxLods, ox, oy);
            s_origin[tid] = int2{ox, oy};
        }
        if (tid < tile_count) s_tiles[tid] = {tid, 0u, 0u, 0u};
    }
    __syncthreads();

    for (uint32_t pass = 0; pass <= view.refinement_count; pass++) {
        const bool from_lds = kAssist && tile_count <= kFrontierCap;  // this pass's parents all sit in s_tiles[pass & 1]
        uint32_t pass_children = 0;                                    // children appended so far in this pass
        // refine_tiles (refine_tiles.wgsl:33-44), 1024 invocation ids per sweep
        for (uint32_t base = 0; base < tile_count; base += kThreads, sweep++) {
            const uint32_t id = base + tid;
            const bool active = id < tile_count;
            bt_tile_coordinate tile{};
            bool divide = false;
            if (active) {
                const int parent_index = (N - 1) * (counter > 0 ? 1 : 0) - int(id) * counter;  // :9-11
                tile = from_lds ? s_tiles[(pass & 1u) * kFrontierCap + id] : temporary_tiles[parent_index];
                if constexpr (kAssist) {
                    const int2 o = s_origin[tile.side * kMaxLods + tile.lod];
                    const int bx = int(tile.x) - o.x, by = int(tile.y) - o.y;
                    if (bx >= 0 && by >= 0 && bx < int(kWinW) && by < int(kWinW) && tile.lod < assist_lods) {
                        const uint32_t b = uint32_t(by) * kWinW + uint32_t(bx);
                        divide = (s_bits[(tile.side * kMaxLods + tile.lod) * kWinWords + (b >> 6)] >> (b & 63u)) & 1ull;
                    } else {
                        divide = should_be_divided(view, tile, approximate_height);  // outside its window: the same function, in place
                    }
                } else {
                    divide = should_be_divided(view, tile, approximate_height);
                }
            }
            const bool fin = active && !divide;
            const unsigned long long ballot_d = __ballot(divide), ballot_f = __ballot(fin);
            if (lane == 0) {
                s_divide[sweep & 1u][wave] = uint32_t(__popcll(ballot_d));
                s_final[sweep & 1u][wave] = uint32_t(__popcll(ballot_f));
            }
            if constexpr (kAssist) lds_barrier();  // (the counts are LDS; this pass reads no global data it wrote)
            else __syncthreads();
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
                    const bt_tile_coordinate child = {tile.side, tile.lod + 1u, (tile.x << 1) + (i & 1u), (tile.y << 1) + ((i >> 1) & 1u)};
                    if (ci >= 0 && ci < N) temporary_tiles[ci] = child;
                    if constexpr (kAssist) {  // the same order (append order = the next pass's id order)
                        const uint32_t li = pass_children + 4u * uint32_t(rank) + i;
                        if (li < kFrontierCap) s_tiles[((pass + 1u) & 1u) * kFrontierCap + li] = child;
                    }
                }
            }
            if (fin) {
                const int fi = final_index + int(before_f + uint32_t(__popcll(ballot_f & below)));
                if (fi < N) final_tiles[fi] = tile;
            }
            pass_children += 4u * total_d;
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
        // orders this pass's child stores before the next pass's parent loads: LDS when the next pass reads its parents there
        if (kAssist && tile_count <= kFrontierCap) lds_barrier();
        else __syncthreads();
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

namespace bt {
bt_ctx* tiling_prepass_ctx(const bt_tiling_prepass* t) { return t ? t->ctx : nullptr; }
}  // namespace bt

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
    if (e == hipSuccess) e = hipMalloc((void**)&t->counters, kCounterWords * sizeof(uint32_t));
    if (e == hipSuccess) e = hipMemsetAsync(t->counters, 0, kCounterWords * sizeof(uint32_t), ctx->stream);
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
    if (t->bits) hipFree(t->bits);
    delete t;
}

namespace {
// How deep can this view refine?  A tile of LOD l divides only within subdivision_distance / 2^l of the view, and nothing is
// closer to the view than its height over the (approximate) surface: beyond l = log2(subdivision_distance / height) no tile
// divides.  An ESTIMATE (f32 on the host, the surface taken as the unit sphere / plane of the mesh transform): it only
// decides how many LODs get their bits up front — anything deeper is evaluated in place, so it can cost time, never change
// the result.
uint32_t estimate_lods(const bt_view_state* view) {
    uint32_t lods = kMaxLods;
    const float* m = view->world_from_local;
    const float* it = view->local_from_world_transpose;
    const float d[3] = {view->world_position[0] - m[9], view->world_position[1] - m[10], view->world_position[2] - m[11]};
    float local[3];
    for (int i = 0; i < 3; i++) local[i] = it[3 * i] * d[0] + it[3 * i + 1] * d[1] + it[3 * i + 2] * d[2];
    const float sx = sqrtf(m[0] * m[0] + m[1] * m[1] + m[2] * m[2]), sy = sqrtf(m[3] * m[3] + m[4] * m[4] + m[5] * m[5]);
    const float height = view->spherical ? (sqrtf(local[0] * local[0] + local[1] * local[1] + local[2] * local[2]) - 1.0f) * std::min(sx, sy) : local[1] * sy;
    const float clearance = 0.5f * fabsf(height - view->approximate_height);
    if (clearance > 0.0f && view->subdivision_distance > 0.0f && std::isfinite(clearance)) {
        const float deepest = log2f(view->subdivision_distance / clearance);
        if (deepest < 30.0f) lods = uint32_t(std::max(0.0f, ceilf(deepest))) + 2u;
    }
    return std::min(std::min(lods, kMaxLods), view->refinement_count + 1u);
}

bt_status prepass_check(bt_tiling_prepass* t, const bt_view_state* view) {
    if (!t || !view) return BT_ERR_INVALID_ARGUMENT;
    if (view->refinement_count > 31 || view->origin_lod > 31) {
        set_error("refinement_count %u / origin_lod %u > 31 (tile x/y are u32)", view->refinement_count, view->origin_lod);
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(t->ctx->device));
    if (!t->bits) {
        BT_HIP(hipMalloc((void**)&t->bits, 6 * size_t(kMaxLods) * kWinWords * sizeof(unsigned long long)));
        BT_HIP(hipMemsetAsync(t->bits, 0, 6 * size_t(kMaxLods) * kWinWords * sizeof(unsigned long long), t->ctx->stream));
        BT_HIP(hipFuncSetAttribute((const void*)tiling_prepass_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, int(6 * kMaxLods * (kWinWords * 8 + 8) + 2 * kFrontierCap * 16)));
    }
    return BT_OK;
}
}  // namespace

}  // extern "C"

namespace bt {
// form 0: two launches — every divide test that can matter (independent, chip-wide), then the ordered schedule over the bits;
//         shallow views (few hundred tiles) take the plain kernel, which saves them the second launch
// form 1: the unordered form — the same SET of final tiles and the same indirect arguments, in whatever order the waves arrive
//         (the reference's own order is the arrival order of its atomics); two chip-wide launches, no pass chain, flat ~10 us
//         whatever the frame; temporary_tiles is not written
// form 2: the plain form — one launch, every divide test evaluated inside the pass that needs it; same list, same order as form 0
bt_status tiling_prepass_enqueue(bt_tiling_prepass* t, const bt_view_state* view, const float* device_height, uint32_t form) {
    if (!t || !view) return BT_ERR_INVALID_ARGUMENT;
    if (view->refinement_count > 31 || view->origin_lod > 31) {
        set_error("refinement_count %u / origin_lod %u > 31 (tile x/y are u32)", view->refinement_count, view->origin_lod);
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(t->ctx->device));
    const uint32_t capacity = std::min(t->capacity, view->geometry_tile_count ? view->geometry_tile_count : t->capacity);
    const uint32_t sides = view->spherical ? 6u : 1u;
    // (with the height on the device the estimate works from the host's copy, one frame old: it decides how many LODs get their
    // bits up front and can only cost time)
    const uint32_t lods = form == 2u ? 0u : estimate_lods(view);
    if (form == 2u || (form == 0u && lods < 12u)) {
        tiling_prepass_kernel<false><<<1, kThreads, 0, t->ctx->stream>>>(*view, capacity, t->temporary_tiles, t->final_tiles, t->indirect, t->counters, nullptr, 0u, device_height);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "tiling_prepass_kernel");
        t->unordered = false;
        return BT_OK;
    }
    if (bt_status s = prepass_check(t, view)) return s;
    if (form == 0u) {
        const size_t lds = size_t(sides) * kMaxLods * (kWinWords * sizeof(unsigned long long) + sizeof(int2)) + 2 * size_t(kFrontierCap) * sizeof(bt_tile_coordinate);
        tiling_divide_bits_kernel<<<sides * lods * kWinChunks, 256, 0, t->ctx->stream>>>(*view, lods, kWinK, t->bits, nullptr, nullptr, device_height);
        tiling_prepass_kernel<true><<<1, kThreads, lds, t->ctx->stream>>>(*view, capacity, t->temporary_tiles, t->final_tiles, t->indirect, t->counters, t->bits, lods, device_height);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return hip_fail(e, "tiling_prepass_kernel");
        t->unordered = false;
        return BT_OK;
    }
    const int radius = t->window ? t->window : kWinK;
    const uint32_t W = 2u * uint32_t(radius) + 1u, chunks = (W * W + 255u) / 256u;
    tiling_divide_bits_kernel<<<sides * lods * chunks, 256, 0, t->ctx->stream>>>(*view, lods, radius, t->bits, t->counters, t->indirect, device_height);
    tiling_collect_kernel<<<sides * lods * chunks, 256, 0, t->ctx->stream>>>(*view, lods, radius, capacity, t->bits, t->final_tiles, t->indirect, t->counters, device_height);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return hip_fail(e, "tiling_collect_kernel");
    t->unordered = true;
    t->unordered_capacity = capacity;
    return BT_OK;
}
}  // namespace bt

extern "C" {

bt_status bt_tiling_prepass_run(bt_tiling_prepass* t, const bt_view_state* view) { return bt::tiling_prepass_enqueue(t, view, nullptr, 0u); }
bt_status bt_tiling_prepass_run_unordered(bt_tiling_prepass* t, const bt_view_state* view) { return bt::tiling_prepass_enqueue(t, view, nullptr, 1u); }
bt_status bt_tiling_prepass_run_plain(bt_tiling_prepass* t, const bt_view_state* view) { return bt::tiling_prepass_enqueue(t, view, nullptr, 2u); }

bt_status bt_tiling_prepass_set_window(bt_tiling_prepass* t, uint32_t radius) {
    if (!t || radius > uint32_t(kWinK)) return BT_ERR_INVALID_ARGUMENT;
    t->window = int(radius);
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
    uint32_t counters[kCounterWords] = {0};
    BT_HIP(hipMemcpyAsync(counters, t->counters, sizeof counters, hipMemcpyDeviceToHost, t->ctx->stream));
    if (indirect) BT_HIP(hipMemcpyAsync(indirect, t->indirect, sizeof(bt_indirect), hipMemcpyDeviceToHost, t->ctx->stream));
    BT_HIP(hipStreamSynchronize(t->ctx->stream));
    if (t->unordered) {  // the reference's buffers: a pass needs its parents + the children appended, the final list its tiles
        counters[1] = counters[0] > t->unordered_capacity;
        for (uint32_t l = 0; l < kMaxLods; l++)
            if (uint64_t(counters[kCntVisited + l]) + 4ull * counters[kCntDivide + l] > t->unordered_capacity) counters[1] = 1;
    }
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
