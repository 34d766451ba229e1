// Host half of libbevy_terrain_amd.so: context, TileCoordinate maths, the TileAtlas index allocator,
// the Preprocessor task queue (the integer tile-index contract) and the tile / config file writers.
// Mirrors the reference's CPU-side behaviour (file:line citations relative to the reference checkout);
// the arithmetic on texels lives in bt_kernels.hip and bt_fused.hip.
#include <dirent.h>
#include <fcntl.h>
#include <sched.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <array>
#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <tuple>

#include "bt_internal.hpp"

namespace bt {

static thread_local char g_error[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof g_error, fmt, ap);
    va_end(ap);
}

bt_status hip_fail(hipError_t e, const char* what) {
    set_error("HIP error %d (%s) in %s", int(e), hipGetErrorString(e), what);
    return BT_ERR_DEVICE;
}

// ------------------------------------------------------------------ TileCoordinate (coordinate.rs)

// coordinate.rs:9-16: for each side: itself, then the side behind its -x, -y, +x, +y edge.
static const uint32_t kNeighbouringSides[6][5] = {
    {0, 4, 2, 1, 5}, {1, 0, 2, 3, 5}, {2, 0, 4, 3, 1}, {3, 2, 4, 5, 1}, {4, 2, 0, 5, 3}, {5, 4, 0, 1, 3},
};

// coordinate.rs:18-53: how a tile position on `side` maps onto `other` (per output axis).
enum Axis : uint8_t { kZero, kLast, kS, kT };
static const Axis kEven[6][2] = {{kS, kT}, {kZero, kT}, {kZero, kS}, {kT, kS}, {kT, kZero}, {kS, kZero}};
static const Axis kOdd[6][2] = {{kS, kT}, {kS, kLast}, {kT, kLast}, {kT, kS}, {kLast, kS}, {kLast, kT}};

void tile_children(bt_tile_coordinate c, bt_tile_coordinate out[4]) {
    for (uint32_t i = 0; i < 4; i++) out[i] = {c.side, c.lod + 1, (c.x << 1) + (i & 1u), (c.y << 1) + (i >> 1)};
}

static bt_tile_coordinate neighbour_at(bt_tile_coordinate c, int nx, int ny, bool spherical) {
    const bt_tile_coordinate invalid = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
    if (c.lod > 30u || (spherical && c.side >= 6u)) return invalid;  // (not a tile: TileCoordinate::INVALID and the like have no neighbours)
    const int n = int(1u << c.lod);
    const bool out_x = nx < 0 || nx >= n, out_y = ny < 0 || ny >= n;
    if (!spherical) {
        if (out_x || out_y) return invalid;
        return {c.side, c.lod, uint32_t(nx), uint32_t(ny)};
    }
    if (out_x && out_y) return invalid;  // cube corner: no diagonal neighbour (coordinate.rs:231-239)
    const int edge = nx < 0 ? 1 : ny < 0 ? 2 : nx >= n ? 3 : ny >= n ? 4 : 0;
    const uint32_t s = uint32_t(std::clamp(nx, 0, n - 1)), t = uint32_t(std::clamp(ny, 0, n - 1));
    const uint32_t other = kNeighbouringSides[c.side][edge];
    const Axis* info = (c.side % 2 == 0 ? kEven : kOdd)[(6 + other - c.side) % 6];
    uint32_t xy[2];
    for (int k = 0; k < 2; k++)
        xy[k] = info[k] == kZero ? 0u : info[k] == kLast ? uint32_t(n - 1) : info[k] == kS ? s : t;
    return {other, c.lod, xy[0], xy[1]};
}

void tile_neighbours(bt_tile_coordinate c, bool spherical, bt_tile_coordinate out[8]) {
    // N, E, S, W, NW, NE, SE, SW (coordinate.rs:209-218) == the region order of stitch.wgsl:57-66
    static const int kOffsets[8][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    for (int i = 0; i < 8; i++) out[i] = neighbour_at(c, int(c.x) + kOffsets[i][0], int(c.y) + kOffsets[i][1], spherical);
}

// ------------------------------------------------------------------ bincode varints (formats/mod.rs)

static uint64_t put_varint(uint64_t v, uint8_t* out, uint64_t pos, uint64_t cap) {
    uint8_t tmp[9];
    uint32_t n = 1;
    if (v < 251) {
        tmp[0] = uint8_t(v);
    } else {
        const uint32_t bytes = v < (1ull << 16) ? 2 : v < (1ull << 32) ? 4 : 8;
        tmp[0] = bytes == 2 ? 251 : bytes == 4 ? 252 : 253;
        for (uint32_t k = 0; k < bytes; k++) tmp[1 + k] = uint8_t(v >> (8 * k));
        n = 1 + bytes;
    }
    if (out && pos + n <= cap) memcpy(out + pos, tmp, n);
    return pos + n;
}

static bool get_varint(const uint8_t* in, uint64_t n, uint64_t& pos, uint64_t& v) {
    if (pos >= n) return false;
    const uint8_t tag = in[pos++];
    if (tag < 251) {
        v = tag;
        return true;
    }
    const uint32_t bytes = tag == 251 ? 2 : tag == 252 ? 4 : tag == 253 ? 8 : 0;
    if (!bytes || pos + bytes > n) return false;
    v = 0;
    for (uint32_t k = 0; k < bytes; k++) v |= uint64_t(in[pos + k]) << (8 * k);
    pos += bytes;
    return true;
}

static bt_status write_file(const std::string& path, const void* data, size_t bytes) {
    FILE* f = fopen(path.c_str(), "wb");
    if (!f) {
        set_error("cannot open %s: %s", path.c_str(), strerror(errno));
        return BT_ERR_IO;
    }
    const size_t wr = fwrite(data, 1, bytes, f);
    fclose(f);
    if (wr != bytes) {
        set_error("short write to %s", path.c_str());
        return BT_ERR_IO;
    }
    return BT_OK;
}

static void remove_tree(const std::string& dir) {
    DIR* d = opendir(dir.c_str());
    if (!d) return;
    while (dirent* e = readdir(d)) {
        if (!strcmp(e->d_name, ".") || !strcmp(e->d_name, "..")) continue;
        const std::string p = dir + "/" + e->d_name;
        struct stat st;
        if (lstat(p.c_str(), &st) == 0 && S_ISDIR(st.st_mode))
            remove_tree(p);
        else
            unlink(p.c_str());
    }
    closedir(d);
    rmdir(dir.c_str());
}

static bt_status make_dirs(const std::string& dir) {
    std::string cur;
    for (size_t i = 0; i <= dir.size(); i++) {
        if (i == dir.size() || dir[i] == '/') {
            if (!cur.empty() && mkdir(cur.c_str(), 0777) != 0 && errno != EEXIST) {
                set_error("mkdir %s: %s", cur.c_str(), strerror(errno));
                return BT_ERR_IO;
            }
        }
        if (i < dir.size()) cur.push_back(dir[i]);
    }
    return BT_OK;
}

// CPUs this process may use: the scheduler's affinity mask, capped by the cgroup's CPU quota (cgroup v2 cpu.max, v1 cfs quota) — a
// container on a 256-thread host may own 16 of them, and std::thread::hardware_concurrency() reports the host's
uint32_t usable_cpus() {
    uint32_t n = 0;
    // (a dynamically sized mask: the fixed cpu_set_t holds 1024 CPUs and sched_getaffinity fails with EINVAL on a larger host)
    for (size_t cpus = 1024; cpus <= (1u << 20) && n == 0; cpus *= 4) {
        cpu_set_t* set = CPU_ALLOC(cpus);
        if (!set) break;
        const size_t bytes = CPU_ALLOC_SIZE(cpus);
        CPU_ZERO_S(bytes, set);
        const int rc = sched_getaffinity(0, bytes, set);
        if (rc == 0) n = uint32_t(CPU_COUNT_S(bytes, set));
        CPU_FREE(set);
        if (rc != 0 && errno != EINVAL) break;
    }
    if (n == 0) n = std::max(1u, std::thread::hardware_concurrency());
    auto quota_from = [](const std::string& path, const std::string& period_path) -> double {
        FILE* f = fopen(path.c_str(), "r");
        if (!f) return 0.0;
        char a[64] = "", b2[64] = "";
        const int got = fscanf(f, "%63s %63s", a, b2);
        fclose(f);
        if (got < 1 || !strcmp(a, "max")) return 0.0;
        double quota = atof(a), period = got >= 2 ? atof(b2) : 0.0;
        if (!period_path.empty()) {
            FILE* g = fopen(period_path.c_str(), "r");
            if (g) {
                if (fscanf(g, "%63s", b2) == 1) period = atof(b2);
                fclose(g);
            }
        }
        return quota > 0.0 && period > 0.0 ? quota / period : 0.0;
    };
    // The process's OWN cgroup (/proc/self/cgroup): without a cgroup namespace — a systemd slice, some Kubernetes set-ups — the quota sits in
    // a nested directory, not at the mount's root; every ancestor's quota applies, the smallest wins.  v2: "0::/path"; v1: "N:cpu,cpuacct:/path".
    std::string v2_path, v1_path;
    if (FILE* f = fopen("/proc/self/cgroup", "r")) {
        char line[1024];
        while (fgets(line, sizeof line, f)) {
            std::string l(line);
            while (!l.empty() && (l.back() == '\n' || l.back() == '\r')) l.pop_back();
            const size_t c1 = l.find(':'), c2 = c1 == std::string::npos ? c1 : l.find(':', c1 + 1);
            if (c2 == std::string::npos) continue;
            const std::string controllers = l.substr(c1 + 1, c2 - c1 - 1), path = l.substr(c2 + 1);
            if (controllers.empty()) v2_path = path;
            else if (("," + controllers + ",").find(",cpu,") != std::string::npos) v1_path = path;
        }
        fclose(f);
    }
    double q = 0.0;
    auto walk = [&](const std::string& root, std::string path, const char* file, const char* period_file) {
        for (;;) {  // the cgroup itself, then every ancestor up to the mount's root
            const std::string dir = root + (path == "/" ? "" : path);
            const double v = quota_from(dir + "/" + file, period_file ? dir + "/" + period_file : std::string());
            if (v > 0.0 && (q <= 0.0 || v < q)) q = v;
            if (path.empty() || path == "/") break;
            const size_t cut = path.find_last_of('/');
            path = cut == 0 || cut == std::string::npos ? "/" : path.substr(0, cut);
        }
    };
    walk("/sys/fs/cgroup", v2_path.empty() ? "/" : v2_path, "cpu.max", nullptr);
    if (q <= 0.0) walk("/sys/fs/cgroup/cpu", v1_path.empty() ? "/" : v1_path, "cpu.cfs_quota_us", "cpu.cfs_period_us");
    if (q > 0.0) n = std::min(n, std::max(1u, uint32_t(q + 0.999)));
    return n;
}

// Writer / reader threads of the save and load paths.  Automatic: min(16, usable CPUs) — on tmpfs and the overlay disk 6 / 8 / 12 / 16 /
// 24 / 32 / 64 threads write at 29 / 34 / 42 / 46-49 / 46 / 35 / 4 GB/s on a 16-CPU quota (beyond ~24 the page-cache allocation lock
// dominates): the threads mostly sleep in the kernel's copy, so the quota itself, not quota - 2, is the optimum there.
uint32_t ctx_io_threads(const bt_ctx* ctx) {
    if (ctx->io_threads) return ctx->io_threads;
    return std::max(1u, std::min(16u, usable_cpus()));
}

bt_status ctx_staging(bt_ctx* ctx, size_t bytes) {
    if (ctx->staging_bytes >= bytes && ctx->staging[0]) return BT_OK;
    for (void*& p : ctx->staging) {
        if (p) hipHostFree(p);
        p = nullptr;
    }
    ctx->staging_bytes = 0;
    for (void*& p : ctx->staging) BT_HIP(hipHostMalloc(&p, bytes, hipHostMallocDefault));
    ctx->staging_bytes = bytes;
    return BT_OK;
}

}  // namespace bt

using namespace bt;

// =====================================================================================  context

extern "C" {

uint32_t bt_abi_version(void) { return BT_ABI_VERSION; }
const char* bt_last_error(void) { return g_error; }

bt_status bt_ctx_create(int32_t device, void* stream, bt_ctx** out) {
    if (!out) return BT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    BT_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) {
        set_error("device %d out of range (%d visible)", device, count);
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(device));
    bt_ctx* ctx = new bt_ctx();
    ctx->device = device;
    if (stream) {
        ctx->stream = hipStream_t(stream);
    } else {
        hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
        if (e != hipSuccess) {
            delete ctx;
            return hip_fail(e, "hipStreamCreateWithFlags");
        }
        ctx->own_stream = true;
    }
    hipEventCreate(&ctx->ev_begin);
    hipEventCreate(&ctx->ev_end);
    *out = ctx;
    return BT_OK;
}

void bt_ctx_destroy(bt_ctx* ctx) {
    if (!ctx) return;
    hipSetDevice(ctx->device);
    for (void* p : ctx->staging)
        if (p) hipHostFree(p);
    if (ctx->ev_begin) hipEventDestroy(ctx->ev_begin);
    if (ctx->ev_end) hipEventDestroy(ctx->ev_end);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    for (auto& kept : ctx->spare_rasters) hipFree(kept.first);
    if (ctx->copy_stream) hipStreamDestroy(ctx->copy_stream);
    if (ctx->save_stream) hipStreamDestroy(ctx->save_stream);
    delete ctx;
}

bt_status bt_ctx_set_stream(bt_ctx* ctx, void* stream) {
    if (!ctx) return BT_ERR_INVALID_ARGUMENT;
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    ctx->own_stream = false;
    ctx->stream = hipStream_t(stream);
    return BT_OK;
}

void* bt_ctx_stream(const bt_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

bt_status bt_ctx_synchronize(bt_ctx* ctx) {
    if (!ctx) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    return BT_OK;
}

bt_status bt_ctx_trim(bt_ctx* ctx, uint64_t* freed_bytes) {
    if (!ctx) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    if (ctx->copy_stream) BT_HIP(hipStreamSynchronize(ctx->copy_stream));
    if (ctx->save_stream) BT_HIP(hipStreamSynchronize(ctx->save_stream));
    uint64_t freed = 0;
    for (auto& kept : ctx->spare_rasters) {
        BT_HIP(hipFree(kept.first));
        freed += kept.second;
    }
    ctx->spare_rasters.clear();
    for (void*& s : ctx->staging)
        if (s) {
            BT_HIP(hipHostFree(s));
            s = nullptr;
            freed += ctx->staging_bytes;
        }
    ctx->staging_bytes = 0;
    if (freed_bytes) *freed_bytes = freed;
    return BT_OK;
}

bt_status bt_ctx_set_io_threads(bt_ctx* ctx, uint32_t threads) {
    if (!ctx || threads > 256u) return BT_ERR_INVALID_ARGUMENT;
    ctx->io_threads = threads;
    return BT_OK;
}

uint32_t bt_ctx_io_threads(const bt_ctx* ctx) { return ctx ? bt::ctx_io_threads(ctx) : 0u; }

bt_status bt_ctx_timer_begin(bt_ctx* ctx) {
    if (!ctx) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipEventRecord(ctx->ev_begin, ctx->stream));
    return BT_OK;
}

bt_status bt_ctx_timer_end(bt_ctx* ctx, float* elapsed_ms) {
    if (!ctx || !elapsed_ms) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipEventRecord(ctx->ev_end, ctx->stream));
    BT_HIP(hipEventSynchronize(ctx->ev_end));
    BT_HIP(hipEventElapsedTime(elapsed_ms, ctx->ev_begin, ctx->ev_end));
    return BT_OK;
}

bt_status bt_device_malloc(bt_ctx* ctx, size_t bytes, void** out) {
    if (!ctx || !out) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipMalloc(out, bytes ? bytes : 1));
    return BT_OK;
}

bt_status bt_device_free(bt_ctx* ctx, void* ptr) {
    if (!ctx) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    BT_HIP(hipFree(ptr));
    return BT_OK;
}

bt_status bt_memcpy_h2d(bt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    return BT_OK;
}

bt_status bt_memcpy_d2h(bt_ctx* ctx, void* dst, const void* src, size_t bytes) {
    if (!ctx) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    BT_HIP(hipStreamSynchronize(ctx->stream));
    return BT_OK;
}

// ==============================================================================  TileCoordinate

void bt_tile_children(bt_tile_coordinate c, bt_tile_coordinate out[4]) {
    if (out) tile_children(c, out);
}
void bt_tile_neighbours(bt_tile_coordinate c, uint32_t spherical, bt_tile_coordinate out[8]) {
    if (out) tile_neighbours(c, spherical != 0, out);
}
bt_tile_coordinate bt_tile_parent(bt_tile_coordinate c) { return {c.side, c.lod - 1u, c.x >> 1, c.y >> 1}; }
int32_t bt_tile_name(bt_tile_coordinate c, char* buf, size_t cap) {
    return snprintf(buf, cap, "%u_%u_%u_%u", c.side, c.lod, c.x, c.y);
}

// ===================================================================================  TileAtlas

bt_status bt_atlas_create(bt_ctx* ctx, const bt_terrain_config* config, bt_atlas** out) {
    if (!ctx || !config || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (config->attachment_count > BT_MAX_ATTACHMENTS) {
        set_error("at most %u attachments", BT_MAX_ATTACHMENTS);
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(ctx->device));
    // everything is checked before anything is allocated, and the device memory comes before the host-side slot list: a nonsensical
    // atlas_size or mip_level_count is an error status, not gigabytes of host memory followed by one
    size_t free_bytes = 0, total_bytes = 0;
    BT_HIP(hipMemGetInfo(&free_bytes, &total_bytes));
    uint64_t wanted = 0;
    for (uint32_t i = 0; i < config->attachment_count; i++) {
        const bt_attachment_config& c = config->attachments[i];
        if (c.texture_size == 0 || c.texture_size > 65536u || c.border_size >= c.texture_size || 2 * c.border_size >= c.texture_size) {
            set_error("attachment %u: texture_size %u / border_size %u", i, c.texture_size, c.border_size);
            return BT_ERR_INVALID_ARGUMENT;
        }
        uint32_t most_mips = 1;
        while ((c.texture_size >> most_mips) != 0) most_mips++;
        if (c.mip_level_count > most_mips) {  // (wgpu refuses such a texture descriptor: gpu_tile_atlas.rs:233-252)
            set_error("attachment %u: mip_level_count %u, a %u-texel texture has at most %u", i, c.mip_level_count, c.texture_size, most_mips);
            return BT_ERR_INVALID_ARGUMENT;
        }
        const uint64_t tile_bytes = uint64_t(c.texture_size) * c.texture_size * (c.format == BT_FORMAT_R16 ? 2u : c.format == BT_FORMAT_RGB8 ? 3u : 4u);
        const bool fits = config->atlas_size <= uint64_t(total_bytes) / tile_bytes;  // (no product that could wrap)
        if (fits) wanted += tile_bytes * config->atlas_size;
        if (!fits || wanted > uint64_t(total_bytes)) {
            set_error("atlas of %u layers does not fit the device (%llu MiB)", config->atlas_size, (unsigned long long)(total_bytes >> 20));
            return BT_ERR_DEVICE;
        }
    }
    bt_atlas* a = new bt_atlas();
    a->ctx = ctx;
    a->config = *config;
    for (uint32_t i = 0; i < config->attachment_count; i++) {
        const bt_attachment_config& c = config->attachments[i];
        Attachment at;
        at.cfg = c;
        // pixel sizes: terrain_data/mod.rs:77-84
        const uint32_t px = c.format == BT_FORMAT_R16 ? 2 : c.format == BT_FORMAT_RGB8 ? 3 : 4;
        at.meta = {c.format, c.texture_size, c.border_size, c.texture_size - 2 * c.border_size, config->atlas_size, px, c.texture_size};
        at.tile_bytes = uint64_t(c.texture_size) * c.texture_size * px;
        const size_t bytes = size_t(at.tile_bytes) * config->atlas_size;
        hipError_t e = hipMalloc(&at.level0, bytes ? bytes : 1);
        if (e == hipSuccess) e = hipMemsetAsync(at.level0, 0, bytes, ctx->stream);  // wgpu textures start zeroed
        if (e != hipSuccess) {
            bt_atlas_destroy(a);
            return hip_fail(e, "atlas allocation");
        }
        at.mips.assign(c.mip_level_count ? c.mip_level_count : 1, nullptr);
        at.written.assign(config->atlas_size, 0);
        a->attachments.push_back(at);
    }
    for (uint32_t i = 0; i < config->atlas_size; i++)  // tile_atlas.rs:307-309
        a->unused_tiles.push_back({{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, i, {0, 0, 0}});
    *out = a;
    return BT_OK;
}

void bt_atlas_destroy(bt_atlas* a) {
    if (!a) return;
    hipSetDevice(a->ctx->device);
    for (Attachment& at : a->attachments) {
        if (at.level0) hipFree(at.level0);
        for (void* m : at.mips)
            if (m) hipFree(m);
    }
    delete a;
}

// tile_atlas.rs:369-381.  (A tile that load_tile_config marked as existing but that was never requested has no
// TileState: the reference's `.unwrap()` panics there; here it reads as INVALID.)
bt_status bt_atlas_get_tile(bt_atlas* a, bt_tile_coordinate c, bt_atlas_tile* out) {
    if (!a || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = {c, BT_INVALID_ATLAS_INDEX, {0, 0, 0}};
    if (is_invalid(c)) return BT_OK;
    if (!a->existing_tiles.count(c)) return BT_OK;
    auto it = a->tile_states.find(c);
    if (it != a->tile_states.end()) out->atlas_index = it->second.atlas_index;
    return BT_OK;
}

// allocate_tile (tile_atlas.rs:383-389): the oldest unused slot; whatever tile was cached in it is forgotten.
// DELIBERATE DEVIATION: called from request_tile the reference has already mem::take()n tile_states, so ITS
// `tile_states.remove(evicted)` is a no-op there and the evicted tile's stale TileState (an atlas_index that now belongs to
// another tile) survives until something overwrites it; get_best_tile / a re-request of the evicted tile then see it.
// Here the evicted tile's state is erased on every path — the sane behaviour the reference's own get_or_allocate path has;
// the oracle (oracle/bt_oracle_tree.c) encodes the same choice.
static bt_status allocate_tile(bt_atlas* a, uint32_t* atlas_index) {
    if (a->unused_tiles.empty()) {
        set_error("Atlas out of indices (atlas_size %u)", a->config.atlas_size);
        return BT_ERR_ATLAS_OUT_OF_INDICES;
    }
    const bt_atlas_tile unused = a->unused_tiles.front();
    a->unused_tiles.pop_front();
    a->tile_states.erase(unused.coordinate);
    a->state_version++;
    *atlas_index = unused.atlas_index;
    return BT_OK;
}

// tile_atlas.rs:391-416
bt_status bt_atlas_get_or_allocate_tile(bt_atlas* a, bt_tile_coordinate c, bt_atlas_tile* out) {
    if (!a || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = {c, BT_INVALID_ATLAS_INDEX, {0, 0, 0}};
    if (is_invalid(c)) return BT_OK;
    auto it = a->tile_states.find(c);
    if (it == a->tile_states.end()) {
        uint32_t index;
        if (bt_status s = allocate_tile(a, &index)) return s;
        it = a->tile_states.emplace(c, TileState{index, 1, 0}).first;
    }
    a->existing_tiles.insert(c);
    out->atlas_index = it->second.atlas_index;
    return BT_OK;
}

// tile_atlas.rs:418-457
bt_status bt_atlas_request_tile(bt_atlas* a, bt_tile_coordinate c) {
    if (!a) return BT_ERR_INVALID_ARGUMENT;
    if (!a->existing_tiles.count(c)) return BT_OK;
    auto it = a->tile_states.find(c);
    if (it != a->tile_states.end()) {
        if (it->second.requests == 0) {  // the tile is used again: take its slot out of the LRU
            const uint32_t index = it->second.atlas_index;
            a->unused_tiles.erase(std::remove_if(a->unused_tiles.begin(), a->unused_tiles.end(),
                                                 [index](const bt_atlas_tile& t) { return t.atlas_index == index; }),
                                  a->unused_tiles.end());
        }
        it->second.requests++;
        return BT_OK;
    }
    uint32_t index;
    if (bt_status s = allocate_tile(a, &index)) return s;
    const uint32_t attachments = uint32_t(a->attachments.size());
    a->tile_states.emplace(c, TileState{index, 1, attachments});
    a->state_version++;
    for (uint32_t ai = 0; ai < attachments; ai++) a->to_load.push_back({c, index, ai});
    if (attachments == 0) a->tile_states[c].loading = 0;  // (Loading(0) never completes upstream; nothing to load here)
    return BT_OK;
}

// tile_atlas.rs:459-476
bt_status bt_atlas_release_tile(bt_atlas* a, bt_tile_coordinate c) {
    if (!a) return BT_ERR_INVALID_ARGUMENT;
    if (!a->existing_tiles.count(c)) return BT_OK;
    auto it = a->tile_states.find(c);
    if (it == a->tile_states.end() || it->second.requests == 0) {
        set_error("Tried releasing a tile, which is not present.");  // the reference panics (:467)
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (--it->second.requests == 0) a->unused_tiles.push_back({c, it->second.atlas_index, {0, 0, 0}});
    return BT_OK;
}

// tile_atlas.rs:478-503
bt_status bt_atlas_get_best_tile(const bt_atlas* a, bt_tile_coordinate c, bt_tile_tree_entry* out) {
    if (!a || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = {BT_INVALID_ATLAS_INDEX, BT_INVALID_LOD};
    while (!is_invalid(c) && c.lod != BT_INVALID_LOD) {
        auto it = a->tile_states.find(c);
        if (it != a->tile_states.end() && it->second.loading == 0) {
            *out = {it->second.atlas_index, c.lod};
            return BT_OK;
        }
        c = bt_tile_parent(c);  // lod 0 -> lod - 1 wraps to INVALID_LOD, like the reference's u32 arithmetic
    }
    return BT_OK;
}

uint32_t bt_atlas_tiles(const bt_atlas* a, bt_tile_coordinate* coords, uint32_t* idx, uint32_t cap) {
    if (!a) return 0;
    std::vector<std::pair<uint32_t, bt_tile_coordinate>> v;
    v.reserve(a->existing_tiles.size());
    for (const bt_tile_coordinate& c : a->existing_tiles) {
        auto it = a->tile_states.find(c);
        v.push_back({it != a->tile_states.end() ? it->second.atlas_index : BT_INVALID_ATLAS_INDEX, c});
    }
    std::sort(v.begin(), v.end(), [](const auto& l, const auto& r) {
        if (l.first != r.first) return l.first < r.first;
        return std::tie(l.second.side, l.second.lod, l.second.x, l.second.y) < std::tie(r.second.side, r.second.lod, r.second.x, r.second.y);
    });
    for (uint32_t i = 0; i < v.size() && i < cap; i++) {
        if (coords) coords[i] = v[i].second;
        if (idx) idx[i] = v[i].first;
    }
    return uint32_t(v.size());
}

bt_status bt_atlas_attachment_storage(const bt_atlas* a, uint32_t ai, void** ptr, uint64_t* tile_bytes, uint32_t* layers) {
    if (!a || ai >= a->attachments.size()) return BT_ERR_INVALID_ARGUMENT;
    if (ptr) {
        *ptr = a->attachments[ai].level0;
        // the caller may write through this pointer (a host-side collective does): no layer counts as "still zero since bt_atlas_create" any more
        const Attachment& at = a->attachments[ai];
        at.mark_written(0, uint32_t(at.written.size()));
    }
    if (tile_bytes) *tile_bytes = a->attachments[ai].tile_bytes;
    if (layers) *layers = a->config.atlas_size;
    return BT_OK;
}

bt_status bt_atlas_download_tiles(bt_atlas* a, uint32_t ai, uint32_t first, uint32_t count, void* dst, uint64_t dst_bytes) {
    if (!a || ai >= a->attachments.size() || !dst) return BT_ERR_INVALID_ARGUMENT;
    const Attachment& at = a->attachments[ai];
    if (uint64_t(first) + count > a->config.atlas_size || dst_bytes < at.tile_bytes * count) {
        set_error("download range/size");
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipMemcpyAsync(dst, (const uint8_t*)at.level0 + at.tile_bytes * first, at.tile_bytes * count,
                          hipMemcpyDeviceToHost, a->ctx->stream));
    BT_HIP(hipStreamSynchronize(a->ctx->stream));
    return BT_OK;
}

bt_status bt_atlas_upload_tile(bt_atlas* a, uint32_t ai, uint32_t layer, const void* src, uint64_t src_bytes) {
    if (!a || ai >= a->attachments.size() || !src) return BT_ERR_INVALID_ARGUMENT;
    const Attachment& at = a->attachments[ai];
    if (layer >= a->config.atlas_size || src_bytes != at.tile_bytes) return BT_ERR_INVALID_ARGUMENT;
    a->attachments[ai].mark_written(layer, 1);
    BT_HIP(hipMemcpyAsync((uint8_t*)at.level0 + at.tile_bytes * layer, src, src_bytes, hipMemcpyHostToDevice, a->ctx->stream));
    BT_HIP(hipStreamSynchronize(a->ctx->stream));
    return BT_OK;
}

}  // extern "C"

namespace {

#ifdef BT_DEBUG_HOOKS
// tools build: BT_STREAM_TRACE=1 prints host time stamps of the streamed run's launcher, its saver thread and the TileSaver underneath
std::chrono::steady_clock::time_point g_trace_start;
bool g_trace = false;
void trace_stamp(const char* what, size_t k) {
    if (g_trace) fprintf(stderr, "[stream] %7.3f ms %s %zu\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - g_trace_start).count(), what, k);
}
#else
inline void trace_stamp(const char*, size_t) {}
#endif

// fs::write for a batch of files on a few threads (the reference spawns one AsyncComputeTaskPool task per tile,
// tile_atlas.rs:77-116): jobs are (path, bytes) pairs; a chunk's pinned buffer is reused once its jobs are done.
class FileWriters {
  public:
    struct Job {
        std::string path;
        const uint8_t* data;
        size_t bytes;
        uint32_t buffer;
        bool read = false;  // fill `data` from the file, which must hold exactly `bytes` (tile load path)
    };
    FileWriters(uint32_t threads, uint32_t buffers) : pending_(buffers, 0) {
        for (uint32_t i = 0; i < threads; i++) workers_.emplace_back([this] { run(); });
    }
    ~FileWriters() {
        {
            std::lock_guard<std::mutex> lock(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (std::thread& t : workers_) t.join();
    }
    void push(std::vector<Job>&& jobs) {
        {
            std::lock_guard<std::mutex> lock(m_);
            for (Job& j : jobs) {
                pending_[j.buffer]++;
                queue_.push_back(std::move(j));
            }
        }
        cv_.notify_all();
    }
    void wait_buffer(uint32_t buffer) {
        std::unique_lock<std::mutex> lock(m_);
        done_.wait(lock, [&] { return pending_[buffer] == 0; });
    }
    bt_status status() {
        std::lock_guard<std::mutex> lock(m_);
        if (failed_) set_error("%s", error_.c_str());
        return failed_ ? BT_ERR_IO : BT_OK;
    }

  private:
    void run() {
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [&] { return stop_ || !queue_.empty(); });
                if (queue_.empty()) return;
                j = std::move(queue_.front());
                queue_.pop_front();
            }
            bool ok = false;
            std::string why;
            if (j.read) {
                const int fd = open(j.path.c_str(), O_RDONLY);
                if (fd >= 0) {
                    uint8_t* dst = const_cast<uint8_t*>(j.data);
                    size_t done = 0;
                    while (done < j.bytes) {
                        const ssize_t r = ::read(fd, dst + done, j.bytes - done);
                        if (r <= 0) break;
                        done += size_t(r);
                    }
                    uint8_t extra;
                    ok = done == j.bytes && ::read(fd, &extra, 1) == 0;
                    close(fd);
                    if (!ok) why = "tile file " + j.path + " does not hold " + std::to_string(j.bytes) + " bytes";
                } else {
                    why = "tile file not found: " + j.path;
                }
                {
                    std::lock_guard<std::mutex> lock(m_);
                    if (!ok && !failed_) {
                        failed_ = true;
                        error_ = why;
                    }
                    pending_[j.buffer]--;
                }
                done_.notify_all();
                continue;
            }
            const int fd = open(j.path.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0666);
            if (fd >= 0) {
                size_t done = 0;
                while (done < j.bytes) {
                    const ssize_t w = write(fd, j.data + done, j.bytes - done);
                    if (w <= 0) break;
                    done += size_t(w);
                }
                ok = done == j.bytes;
                ok = (close(fd) == 0) && ok;
                if (!ok) why = "short write to " + j.path;
            } else {
                why = "cannot open " + j.path + ": " + strerror(errno);
            }
            {
                std::lock_guard<std::mutex> lock(m_);
                if (!ok && !failed_) {
                    failed_ = true;
                    error_ = why;
                }
                pending_[j.buffer]--;
            }
            done_.notify_all();
        }
    }
    std::mutex m_;
    std::condition_variable cv_, done_;
    std::deque<Job> queue_;
    std::vector<uint32_t> pending_;
    std::vector<std::thread> workers_;
    bool stop_ = false, failed_ = false;
    std::string error_;
};

// Download + write tiles: D2H through three pinned buffers on `stream` (runs of consecutive layers are one copy), files written by
// the writer threads while the next chunk downloads.  add() may be called many times (the streamed run hands over band after band,
// attachment after attachment); the tiles of one add() are written in atlas-index order.
class TileSaver {
  public:
    typedef std::vector<std::pair<uint32_t, bt_tile_coordinate>> Tiles;
    TileSaver(bt_atlas* a, hipStream_t stream) : a_(a), stream_(stream) {}
    ~TileSaver() {
        if (writers_)
            for (uint32_t k = 0; k < kBuffers; k++) writers_->wait_buffer(k);
        for (uint32_t k = 0; k < kBuffers; k++)
            if (copied_[k]) hipEventDestroy(copied_[k]);
    }
    bt_status begin() {
        BT_HIP(hipSetDevice(a_->ctx->device));
        size_t largest = 32ull << 20;
        for (const Attachment& at : a_->attachments) largest = std::max<size_t>(largest, at.tile_bytes);
        if (bt_status s = ctx_staging(a_->ctx, largest)) return s;
        for (uint32_t k = 0; k < kBuffers; k++) {
            hipError_t e = hipEventCreateWithFlags(&copied_[k], hipEventDisableTiming);
            if (e != hipSuccess) return hip_fail(e, "save events");
        }
        // 16 writers: measured on tmpfs and the overlay disk, 6 / 8 / 12 / 16 / 24 / 32 / 64 / 128 threads write at 29 / 34 / 42 /
        // 46-49 / 46 / 35 / 4 / 5 GB/s — beyond ~24 the page-cache allocation lock dominates
        // (the count follows the CPUs the process may use, not the machine's hardware threads: bt_ctx_set_io_threads)
        uint32_t threads = ctx_io_threads(a_->ctx);
#ifdef BT_DEBUG_HOOKS
        if (const char* e = getenv("BT_SAVE_THREADS")) threads = std::max(1, atoi(e));  // tools build only: writer-count experiments
#endif
        writers_.reset(new FileWriters(threads, kBuffers));
        return BT_OK;
    }
    // taper: the call's last tiles travel in shrinking chunks (half of what is left, down to 8 tiles) — full-size chunks keep the
    // copy engine at its rate, the small ones at the very end shorten the writers' tail behind the last copy
    bt_status add(uint32_t ai, const std::string& dir, Tiles tiles, bool taper = false) {
        const Attachment& at = a_->attachments[ai];
        void** pinned = a_->ctx->staging;
        if (std::find(dirs_.begin(), dirs_.end(), dir) == dirs_.end()) {
            if (bt_status s = make_dirs(dir)) return s;
            dirs_.push_back(dir);
        }
        const uint32_t chunk = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(64, (32ull << 20) / at.tile_bytes)));
        std::sort(tiles.begin(), tiles.end(), [](const auto& l, const auto& r) { return l.first < r.first; });
        tiles.erase(std::unique(tiles.begin(), tiles.end(), [](const auto& l, const auto& r) { return l.first == r.first && operator_eq(l.second, r.second); }),
                    tiles.end());
        const size_t n = tiles.size();
        for (size_t lo = 0, step = 0; lo < n; lo += step) {
            step = chunk;
            if (taper && n - lo <= 2 * size_t(chunk)) step = std::max<size_t>(std::min<size_t>(8, chunk), (n - lo) / 2);
            const size_t hi = std::min(n, lo + step);
            const uint32_t k = uint32_t(chunks_++ % kBuffers);
            trace_stamp("  saver chunk: tiles", hi - lo);
            writers_->wait_buffer(k);
            trace_stamp("  saver chunk: buffer free", k);
            // runs of consecutive layers; equally long runs at a constant layer stride (a band of tile rows in the x-major
            // atlas order: 4 layers every 32) travel as ONE pitched copy instead of one call per run
            std::vector<std::pair<size_t, size_t>> runs;  // (first tile of the chunk, length)
            for (size_t i = lo; i < hi;) {
                size_t run = 1;
                while (i + run < hi && tiles[i + run].first == tiles[i].first + run) run++;
                runs.push_back({i, run});
                i += run;
            }
            bool regular = runs.size() >= 2;
            const uint64_t stride = regular ? uint64_t(tiles[runs[1].first].first) - tiles[runs[0].first].first : 0;
            for (size_t q = 1; regular && q < runs.size(); q++)
                regular = runs[q].second == runs[0].second && uint64_t(tiles[runs[q].first].first) - tiles[runs[q - 1].first].first == stride;
            if (regular) {
                hipError_t e = hipMemcpy2DAsync(pinned[k], at.tile_bytes * runs[0].second, (const uint8_t*)at.level0 + at.tile_bytes * tiles[lo].first,
                                                at.tile_bytes * stride, at.tile_bytes * runs[0].second, runs.size(), hipMemcpyDeviceToHost, stream_);
                if (e != hipSuccess) return hip_fail(e, "tile download");
            } else if (runs.size() > 2 && hi - lo <= 64 && at.tile_bytes % 16u == 0) {
                // an irregular chunk (the lower LODs behind a band's tiles, a cube's face-edge tiles, merged hand-overs): ONE gather kernel that
                // writes the pinned buffer over PCIe instead of one copy-engine call per run — a burst of small copy calls stalled the issuing
                // thread for 15 - 20 ms now and then (config 2's 37-tile chunk: 2.2 -> 21.0 ms between two stamps, round 6 traces), the kernel never
                uint32_t layers[64];
                for (size_t i = lo; i < hi; i++) layers[i - lo] = tiles[i].first;
                if (bt_status s = launch_gather_layers(stream_, at.level0, layers, uint32_t(hi - lo), pinned[k], at.tile_bytes)) return s;
            } else {
                for (const auto& [i, run] : runs) {
                    hipError_t e = hipMemcpyAsync((uint8_t*)pinned[k] + at.tile_bytes * (i - lo), (const uint8_t*)at.level0 + at.tile_bytes * tiles[i].first,
                                                  at.tile_bytes * run, hipMemcpyDeviceToHost, stream_);
                    if (e != hipSuccess) return hip_fail(e, "tile download");
                }
            }
            hipError_t e = hipEventRecord(copied_[k], stream_);
            if (e != hipSuccess) return hip_fail(e, "tile download");
            trace_stamp("  saver chunk: copy issued", k);
            if (bt_status s = hand_over()) return s;  // the chunk enqueued BEFORE this one: wait for its copies, queue its files
            trace_stamp("  saver chunk: previous chunk handed over", k);
            in_flight_.assign(tiles.begin() + lo, tiles.begin() + hi);
            in_flight_buffer_ = k;
            in_flight_ai_ = ai;
            in_flight_dir_ = dir;
            have_in_flight_ = true;
            saved_bytes_ += uint64_t(hi - lo) * at.tile_bytes;
        }
        return BT_OK;
    }
    bt_status finish() {
        if (bt_status s = hand_over()) return s;
        for (uint32_t k = 0; k < kBuffers; k++) writers_->wait_buffer(k);
        return writers_->status();
    }
    uint64_t saved_bytes() const { return saved_bytes_; }

  private:
    static constexpr uint32_t kBuffers = bt_ctx::kStagingBuffers;
    bt_status hand_over() {
        if (!have_in_flight_) return BT_OK;
        have_in_flight_ = false;
        const Attachment& at = a_->attachments[in_flight_ai_];
        hipError_t e = hipEventSynchronize(copied_[in_flight_buffer_]);
        if (e != hipSuccess) return hip_fail(e, "tile download");
        std::vector<FileWriters::Job> jobs;
        for (size_t i = 0; i < in_flight_.size(); i++) {
            char name[64];
            bt_tile_name(in_flight_[i].second, name, sizeof name);
            jobs.push_back({in_flight_dir_ + "/" + name + ".bin", (const uint8_t*)a_->ctx->staging[in_flight_buffer_] + at.tile_bytes * i, size_t(at.tile_bytes), in_flight_buffer_});
        }
        writers_->push(std::move(jobs));
        return BT_OK;
    }
    bt_atlas* a_;
    hipStream_t stream_;
    size_t chunks_ = 0;
    hipEvent_t copied_[kBuffers] = {};
    std::unique_ptr<FileWriters> writers_;
    std::vector<std::string> dirs_;  // directories that exist by now
    Tiles in_flight_;
    uint32_t in_flight_buffer_ = 0, in_flight_ai_ = 0;
    std::string in_flight_dir_;
    bool have_in_flight_ = false;
    uint64_t saved_bytes_ = 0;
};

bt_status save_tiles(bt_atlas* a, uint32_t ai, const char* directory, std::vector<std::pair<uint32_t, bt_tile_coordinate>> tiles) {
    if (tiles.empty()) return make_dirs(directory);
    TileSaver saver(a, a->ctx->stream);
    if (bt_status s = saver.begin()) return s;
    if (bt_status s = saver.add(ai, directory, std::move(tiles))) return s;
    return saver.finish();
}

}  // namespace

extern "C" {

// every existing tile that holds a slot (a tile known only from load_tile_config has no data to write)
bt_status bt_atlas_save_attachment(bt_atlas* a, uint32_t ai, const char* directory) {
    if (!a || ai >= a->attachments.size() || !directory) return BT_ERR_INVALID_ARGUMENT;
    std::vector<std::pair<uint32_t, bt_tile_coordinate>> tiles;
    for (const bt_tile_coordinate& c : a->existing_tiles) {
        auto it = a->tile_states.find(c);
        if (it != a->tile_states.end() && it->second.atlas_index != BT_INVALID_ATLAS_INDEX) tiles.push_back({it->second.atlas_index, c});
    }
    return save_tiles(a, ai, directory, std::move(tiles));
}

uint64_t bt_tc_encode(const bt_tile_coordinate* tiles, uint32_t count, uint8_t* out, uint64_t cap) {
    uint64_t pos = put_varint(count, out, 0, cap);
    for (uint32_t i = 0; i < count; i++) {
        pos = put_varint(tiles[i].side, out, pos, cap);
        pos = put_varint(tiles[i].lod, out, pos, cap);
        pos = put_varint(tiles[i].x, out, pos, cap);
        pos = put_varint(tiles[i].y, out, pos, cap);
    }
    return pos;
}

int64_t bt_tc_decode(const uint8_t* data, uint64_t bytes, bt_tile_coordinate* tiles, uint32_t cap) {
    uint64_t pos = 0, len = 0;
    if (!data || !get_varint(data, bytes, pos, len)) return -1;
    for (uint64_t i = 0; i < len; i++) {
        uint64_t v[4];
        for (int k = 0; k < 4; k++)
            if (!get_varint(data, bytes, pos, v[k]) || v[k] > 0xFFFFFFFFull) return -1;
        if (tiles && i < cap) tiles[i] = {uint32_t(v[0]), uint32_t(v[1]), uint32_t(v[2]), uint32_t(v[3])};
    }
    return int64_t(len);
}

bt_status bt_atlas_save_tile_config(const bt_atlas* a, const char* file_path) {
    if (!a || !file_path) return BT_ERR_INVALID_ARGUMENT;
    std::vector<bt_tile_coordinate> coords(a->existing_tiles.begin(), a->existing_tiles.end());
    // the reference writes HashSet iteration order (tile_atlas.rs:605-609); sorted here, compare as a set
    std::sort(coords.begin(), coords.end(), [](const bt_tile_coordinate& l, const bt_tile_coordinate& r) {
        return std::tie(l.side, l.lod, l.x, l.y) < std::tie(r.side, r.lod, r.x, r.y);
    });
    std::vector<uint8_t> buf(bt_tc_encode(coords.data(), uint32_t(coords.size()), nullptr, 0));
    bt_tc_encode(coords.data(), uint32_t(coords.size()), buf.data(), buf.size());
    return write_file(file_path, buf.data(), buf.size());
}

bt_status bt_atlas_load_tile_config(bt_atlas* a, const char* file_path) {
    if (!a || !file_path) return BT_ERR_INVALID_ARGUMENT;
    FILE* f = fopen(file_path, "rb");
    if (!f) {
        set_error("Tile config not found: %s", file_path);  // tile_atlas.rs:620
        return BT_ERR_IO;
    }
    std::vector<uint8_t> buf;
    uint8_t tmp[65536];
    size_t r;
    while ((r = fread(tmp, 1, sizeof tmp, f)) > 0) buf.insert(buf.end(), tmp, tmp + r);
    fclose(f);
    const int64_t n = bt_tc_decode(buf.data(), buf.size(), nullptr, 0);
    if (n < 0) {
        set_error("malformed tile config %s", file_path);
        return BT_ERR_IO;
    }
    std::vector<bt_tile_coordinate> coords(size_t(n) ? size_t(n) : 1);
    bt_tc_decode(buf.data(), buf.size(), coords.data(), uint32_t(n));
    // existing_tiles only: a tile gets an atlas index when it is requested or preprocessed
    for (int64_t i = 0; i < n; i++) a->existing_tiles.insert(coords[i]);
    return BT_OK;
}

// ------------------------------------------------------------------------------------- mip chain

static uint32_t mip_pixel_size(uint32_t format) { return format == BT_FORMAT_R16 ? 2 : 4; }

bt_status bt_generate_mipmaps(bt_ctx* ctx, uint32_t format, uint32_t T, uint32_t levels, const void* level0, void* out,
                              uint64_t out_bytes) {
    if (!ctx || !level0 || !out || levels == 0) return BT_ERR_INVALID_ARGUMENT;
    if (format != BT_FORMAT_R16 && format != BT_FORMAT_RGBA8) {
        set_error("generate_mipmaps: format %u is a no-op in the reference (terrain_data/mod.rs:211-213)", format);
        return BT_ERR_UNSUPPORTED;
    }
    const uint32_t px = mip_pixel_size(format);
    uint64_t total = 0;
    for (uint32_t k = 0; k < levels; k++) total += uint64_t(T >> k) * (T >> k) * px;
    if (out_bytes < total) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    uint8_t* dev = nullptr;
    BT_HIP(hipMalloc((void**)&dev, total));
    bt_status rc = BT_OK;
    hipError_t e = hipMemcpyAsync(dev, level0, uint64_t(T) * T * px, hipMemcpyHostToDevice, ctx->stream);
    uint64_t off = 0;
    uint32_t size = T;
    for (uint32_t k = 1; k < levels && e == hipSuccess && rc == BT_OK; k++) {
        const uint64_t next = off + uint64_t(size) * size * px;
        rc = launch_mip_level(ctx, format, dev + off, dev + next, size, 1);
        off = next;
        size >>= 1;
    }
    if (e == hipSuccess && rc == BT_OK) e = hipMemcpyAsync(out, dev, total, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == BT_OK) e = hipStreamSynchronize(ctx->stream);
    hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "bt_generate_mipmaps");
    return rc;
}

bt_status bt_atlas_generate_mipmaps(bt_atlas* a, uint32_t ai, uint32_t first, uint32_t count) {
    if (!a || ai >= a->attachments.size()) return BT_ERR_INVALID_ARGUMENT;
    Attachment& at = a->attachments[ai];
    if (at.meta.format != BT_FORMAT_R16 && at.meta.format != BT_FORMAT_RGBA8) return BT_ERR_UNSUPPORTED;
    if (uint64_t(first) + count > a->config.atlas_size) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(a->ctx->device));
    const uint32_t px = at.meta.pixel_size;
    uint32_t size = at.meta.texture_size;
    const uint8_t* parent = (const uint8_t*)at.level0 + at.tile_bytes * first;
    for (uint32_t k = 1; k < at.mips.size(); k++) {
        const uint32_t child = size >> 1;
        const uint64_t child_tile = uint64_t(child) * child * px;
        if (!at.mips[k]) {
            BT_HIP(hipMalloc(&at.mips[k], child_tile * a->config.atlas_size ? child_tile * a->config.atlas_size : 1));
            BT_HIP(hipMemsetAsync(at.mips[k], 0, child_tile * a->config.atlas_size, a->ctx->stream));
        }
        uint8_t* dst = (uint8_t*)at.mips[k] + child_tile * first;
        if (bt_status s = launch_mip_level(a->ctx, at.meta.format, parent, dst, size, count)) return s;
        parent = dst;
        size = child;
    }
    return BT_OK;
}

// AtlasTileAttachmentWithData::start_loading (tile_atlas.rs:118-149) + GpuAtlasAttachment::upload_tiles
// (gpu_tile_atlas.rs:309-336) for a batch of tiles: "{directory}/{coord}.bin" -> atlas layer of the tile (allocated
// on demand, like request_tile does), then ONE batched mip-chain pass per run of consecutive layers instead of the
// reference's per-tile CPU loop.  coords == NULL: every existing tile (load_tile_config) of the atlas.
bt_status bt_atlas_load_tiles(bt_atlas* a, uint32_t ai, const char* directory, const bt_tile_coordinate* coords, uint32_t count) {
    if (!a || ai >= a->attachments.size() || !directory || (count && !coords)) return BT_ERR_INVALID_ARGUMENT;
    Attachment& at = a->attachments[ai];
    std::vector<bt_tile_coordinate> all;
    if (!coords) {
        for (const bt_tile_coordinate& c : a->existing_tiles) all.push_back(c);
        std::sort(all.begin(), all.end(), [](const bt_tile_coordinate& x, const bt_tile_coordinate& y) {
            return std::tie(x.side, x.lod, x.x, x.y) < std::tie(y.side, y.lod, y.x, y.y);
        });
        coords = all.data();
        count = uint32_t(all.size());
    }
    if (!count) return BT_OK;
    BT_HIP(hipSetDevice(a->ctx->device));
    // files -> three pinned buffers (reader threads) -> H2D on the context's stream: chunk c + 1 is read while chunk c uploads
    constexpr uint32_t kBuffers = bt_ctx::kStagingBuffers;
    const uint32_t chunk = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(64, (32ull << 20) / at.tile_bytes)));
    if (bt_status s = ctx_staging(a->ctx, std::max<size_t>(32ull << 20, at.tile_bytes))) return s;
    void** pinned = a->ctx->staging;
    std::vector<uint32_t> layers;
    hipEvent_t uploaded[kBuffers] = {};
    bool in_flight[kBuffers] = {};
    bt_status rc = BT_OK;
    for (uint32_t k = 0; k < kBuffers && rc == BT_OK; k++) {
        hipError_t e = hipEventCreateWithFlags(&uploaded[k], hipEventDisableTiming);
        if (e != hipSuccess) rc = hip_fail(e, "load events");
    }
    if (rc == BT_OK) {
        FileWriters readers(ctx_io_threads(a->ctx), kBuffers);
        const uint32_t chunks = (count + chunk - 1) / chunk;
        std::vector<std::vector<uint32_t>> index(kBuffers);
        auto upload = [&](uint32_t c) -> bt_status {  // chunk c has been queued for reading: wait for its files, enqueue its copies
            const uint32_t k = c % kBuffers;
            readers.wait_buffer(k);
            if (bt_status s = readers.status()) return s;
            for (size_t t = 0; t < index[k].size();) {  // runs of consecutive layers are one copy
                size_t run = 1;
                while (t + run < index[k].size() && index[k][t + run] == index[k][t] + run) run++;
                a->attachments[ai].mark_written(index[k][t], uint32_t(run));
                hipError_t e = hipMemcpyAsync((uint8_t*)at.level0 + at.tile_bytes * index[k][t], (const uint8_t*)pinned[k] + at.tile_bytes * t,
                                              at.tile_bytes * run, hipMemcpyHostToDevice, a->ctx->stream);
                if (e != hipSuccess) return hip_fail(e, "tile upload");
                t += run;
            }
            hipError_t e = hipEventRecord(uploaded[k], a->ctx->stream);
            if (e != hipSuccess) return hip_fail(e, "tile upload");
            in_flight[k] = true;
            return BT_OK;
        };
        for (uint32_t c = 0; c < chunks && rc == BT_OK; c++) {
            const uint32_t k = c % kBuffers, first = c * chunk, n = std::min(chunk, count - first);
            if (in_flight[k]) {  // the buffer's previous upload must have left it
                hipError_t e = hipEventSynchronize(uploaded[k]);
                if (e != hipSuccess) rc = hip_fail(e, "tile upload");
                in_flight[k] = false;
            }
            std::vector<FileWriters::Job> jobs;
            index[k].clear();
            for (uint32_t t = 0; t < n && rc == BT_OK; t++) {
                bt_atlas_tile tile;
                rc = bt_atlas_get_or_allocate_tile(a, coords[first + t], &tile);
                if (rc) break;
                index[k].push_back(tile.atlas_index);
                layers.push_back(tile.atlas_index);
                char name[64];
                bt_tile_name(coords[first + t], name, sizeof name);
                FileWriters::Job job{std::string(directory) + "/" + name + ".bin", (const uint8_t*)pinned[k] + at.tile_bytes * t, size_t(at.tile_bytes), k};
                job.read = true;
                jobs.push_back(std::move(job));
            }
            if (rc == BT_OK) readers.push(std::move(jobs));
            if (rc == BT_OK && c > 0) rc = upload(c - 1);
        }
        if (rc == BT_OK) rc = upload(chunks - 1);
        for (uint32_t k = 0; k < kBuffers; k++) readers.wait_buffer(k);  // (error paths: nobody may still write into the buffers)
        hipError_t e = hipStreamSynchronize(a->ctx->stream);  // the staging buffers belong to the context: free for the next call
        if (rc == BT_OK && e != hipSuccess) rc = hip_fail(e, "tile upload");
    }
    for (uint32_t k = 0; k < kBuffers; k++)
        if (uploaded[k]) hipEventDestroy(uploaded[k]);
    if (rc || at.mips.size() <= 1) return rc;
    std::sort(layers.begin(), layers.end());
    for (size_t i = 0; i < layers.size();) {
        size_t j = i + 1;
        while (j < layers.size() && layers[j] <= layers[j - 1] + 1) j++;
        if (bt_status s = bt_atlas_generate_mipmaps(a, ai, layers[i], layers[j - 1] - layers[i] + 1)) return s;
        i = j;
    }
    return BT_OK;
}

uint32_t bt_atlas_pending_loads(const bt_atlas* a) { return a ? uint32_t(a->to_load.size()) : 0u; }

// TileAtlasState::update (tile_atlas.rs:327-345: start up to load_slots loads) + AtlasAttachment::update (:195-224:
// finished loads -> loaded_tile_attachment, data into the atlas) + GpuAtlasAttachment::upload_tiles
// (gpu_tile_atlas.rs:309-336), done synchronously for up to `max_loads` queued entries.
bt_status bt_atlas_update(bt_atlas* a, const char* assets_root, uint32_t max_loads, uint32_t* loaded, uint32_t* failed) {
    if (!a || !assets_root) return BT_ERR_INVALID_ARGUMENT;
    if (loaded) *loaded = 0;
    if (failed) *failed = 0;
    if (a->to_load.empty()) return BT_OK;
    BT_HIP(hipSetDevice(a->ctx->device));
    uint64_t largest = 0;
    for (const Attachment& at : a->attachments) largest = std::max(largest, at.tile_bytes);
    const uint32_t chunk = uint32_t(std::max<uint64_t>(1, std::min<uint64_t>(64, (32ull << 20) / std::max<uint64_t>(largest, 1))));
    if (bt_status s = ctx_staging(a->ctx, std::max<size_t>(32ull << 20, largest))) return s;
    void* pinned = a->ctx->staging[0];
    std::vector<std::vector<uint32_t>> mip_layers(a->attachments.size());
    uint32_t budget = max_loads ? max_loads : 0xFFFFFFFFu, done = 0, bad = 0;
    bt_status rc = BT_OK;
    while (!a->to_load.empty() && budget && rc == BT_OK) {
        std::vector<AtlasTileAttachment> batch;
        while (!a->to_load.empty() && budget && batch.size() < chunk) {
            batch.push_back(a->to_load.front());
            a->to_load.pop_front();
            budget--;
        }
        std::vector<bool> ok(batch.size(), false);
        for (size_t k = 0; k < batch.size(); k++) {
            const AtlasTileAttachment& t = batch[k];
            const Attachment& at = a->attachments[t.attachment_index];
            char name[64];
            bt_tile_name(t.coordinate, name, sizeof name);
            const std::string path = std::string(assets_root) + "/" + a->config.path + "/data/" + at.cfg.name + "/" + name + ".bin";
            FILE* f = fopen(path.c_str(), "rb");
            if (!f) continue;  // the reference returns the load slot and leaves the tile Loading (:202-204)
            const size_t got = fread((uint8_t*)pinned + largest * k, 1, at.tile_bytes, f);
            const bool longer = got == at.tile_bytes && fgetc(f) != EOF;
            fclose(f);
            ok[k] = got == at.tile_bytes && !longer;
        }
        for (size_t k = 0; k < batch.size() && rc == BT_OK; k++) {
            if (!ok[k]) {
                bad++;
                continue;
            }
            const AtlasTileAttachment& t = batch[k];
            Attachment& at = a->attachments[t.attachment_index];
            at.mark_written(t.atlas_index, 1);
            hipError_t e = hipMemcpyAsync((uint8_t*)at.level0 + at.tile_bytes * t.atlas_index, (const uint8_t*)pinned + largest * k, at.tile_bytes,
                                          hipMemcpyHostToDevice, a->ctx->stream);
            if (e != hipSuccess) rc = hip_fail(e, "tile upload");
        }
        if (rc == BT_OK) {
            hipError_t e = hipStreamSynchronize(a->ctx->stream);  // the pinned buffer is refilled next
            if (e != hipSuccess) rc = hip_fail(e, "tile upload");
        }
        for (size_t k = 0; k < batch.size() && rc == BT_OK; k++) {
            if (!ok[k]) continue;
            const AtlasTileAttachment& t = batch[k];
            // loaded_tile_attachment (:347-359); a slot that was reused for another tile meanwhile is ignored
            auto it = a->tile_states.find(t.coordinate);
            if (it == a->tile_states.end() || it->second.atlas_index != t.atlas_index || it->second.loading == 0) continue;
            if (--it->second.loading == 0) a->state_version++;
            mip_layers[t.attachment_index].push_back(t.atlas_index);
            done++;
        }
    }
    for (uint32_t ai = 0; ai < a->attachments.size() && rc == BT_OK; ai++) {
        std::vector<uint32_t>& layers = mip_layers[ai];
        if (layers.empty() || a->attachments[ai].mips.size() <= 1) continue;
        std::sort(layers.begin(), layers.end());
        for (size_t i = 0; i < layers.size() && rc == BT_OK;) {
            size_t j = i + 1;
            while (j < layers.size() && layers[j] <= layers[j - 1] + 1) j++;
            rc = bt_atlas_generate_mipmaps(a, ai, layers[i], layers[j - 1] - layers[i] + 1);
            i = j;
        }
    }
    if (loaded) *loaded = done;
    if (failed) *failed = bad;
    return rc;
}

bt_status bt_atlas_sample(bt_atlas* a, uint32_t ai, const bt_tile_lookup* lookups, uint32_t count, float* out) {
    if (!a || ai >= a->attachments.size() || (count && (!lookups || !out))) return BT_ERR_INVALID_ARGUMENT;
    const Attachment& at = a->attachments[ai];
    if (at.meta.format != BT_FORMAT_R16 && at.meta.format != BT_FORMAT_RGBA8) return BT_ERR_UNSUPPORTED;
    if (!count) return BT_OK;
    BT_HIP(hipSetDevice(a->ctx->device));
    void* dev = nullptr;
    const size_t in_bytes = sizeof(bt_tile_lookup) * size_t(count), out_bytes = 16 * size_t(count);
    BT_HIP(hipMalloc(&dev, in_bytes + out_bytes));
    hipError_t e = hipMemcpyAsync(dev, lookups, in_bytes, hipMemcpyHostToDevice, a->ctx->stream);
    bt_status rc = BT_OK;
    if (e == hipSuccess) rc = launch_sample(a->ctx, at.meta, at.level0, (const bt_tile_lookup*)dev, count, (float*)((uint8_t*)dev + in_bytes));
    if (e == hipSuccess && rc == BT_OK) e = hipMemcpyAsync(out, (uint8_t*)dev + in_bytes, out_bytes, hipMemcpyDeviceToHost, a->ctx->stream);
    if (e == hipSuccess && rc == BT_OK) e = hipStreamSynchronize(a->ctx->stream);
    hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "bt_atlas_sample");
    return rc;
}

bt_status bt_atlas_mip_storage(const bt_atlas* a, uint32_t ai, uint32_t level, void** ptr, uint64_t* tile_bytes) {
    if (!a || ai >= a->attachments.size() || level >= a->attachments[ai].mips.size()) return BT_ERR_INVALID_ARGUMENT;
    const Attachment& at = a->attachments[ai];
    const uint32_t s = at.meta.texture_size >> level;
    if (ptr) *ptr = level == 0 ? at.level0 : at.mips[level];
    if (tile_bytes) *tile_bytes = uint64_t(s) * s * at.meta.pixel_size;
    return BT_OK;
}

bt_status bt_synth_fbm_r16(bt_ctx* ctx, void* dst, uint32_t w, uint32_t h, uint64_t pitch, uint32_t x0, uint32_t y0,
                           uint32_t base_cell, uint32_t octaves, uint32_t seed) {
    if (!ctx || !dst || !w || !h || !base_cell || !octaves || octaves > 16) return BT_ERR_INVALID_ARGUMENT;
    BT_HIP(hipSetDevice(ctx->device));
    return launch_synth_fbm(ctx, dst, w, h, pitch ? pitch : uint64_t(w) * 2, x0, y0, base_cell, octaves, seed);
}

// ================================================================================  Preprocessor

bt_status bt_preprocessor_create(bt_ctx* ctx, bt_preprocessor** out) {
    if (!ctx || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = new bt_preprocessor();
    (*out)->ctx = ctx;
    return BT_OK;
}

static void release_rasters(bt_preprocessor* p) {
    // kernels of an asynchronous run may still read these buffers; the hipFree this parking replaces synchronised implicitly
    bool synced = false;
    for (Raster& r : p->rasters)
        if (r.owned && r.dev.data && !synced) {
            hipStreamSynchronize(p->ctx->stream);
            synced = true;
        }
    for (Raster& r : p->rasters)
        if (r.owned && r.dev.data) p->ctx->park_raster((void*)r.dev.data, r.alloc_bytes);  // kept for the next queue's rasters (bt_ctx::spare_rasters)
    p->rasters.clear();
}

void bt_preprocessor_destroy(bt_preprocessor* p) {
    if (!p) return;
    hipSetDevice(p->ctx->device);
    release_rasters(p);
    fused_release(p);
    for (hipEvent_t e : p->events) hipEventDestroy(e);
    for (hipEvent_t e : p->event_pool) hipEventDestroy(e);
    if (p->shard_local_done) hipEventDestroy(p->shard_local_done);
    if (p->shard_exchange_done) hipEventDestroy(p->shard_exchange_done);
    if (p->tasks_dev) hipFree(p->tasks_dev);
    if (p->rasters_dev) hipFree(p->rasters_dev);
    delete p;
}

bt_status bt_preprocessor_clear_attachment(bt_preprocessor* p, bt_atlas* a, uint32_t ai, const char* directory) {
    if (!p || !a || ai >= a->attachments.size()) return BT_ERR_INVALID_ARGUMENT;
    a->existing_tiles.clear();  // for ALL attachments, like :292; the slots (tile_states) stay assigned
    if (directory) {
        // reset_directory (preprocessor.rs:18-22)
        unlink((std::string(directory) + "/../../config.tc").c_str());
        remove_tree(directory);
        return make_dirs(directory);
    }
    return BT_OK;
}

}  // extern "C"

namespace {

struct TileRange {
    uint32_t lx, ly, ux, uy;
};

// PreprocessDataset::overlapping_tiles (preprocessor.rs:58-66): f32 maths; `as_uvec2` saturates (negative -> 0).
// The range is clamped to the face (tiles x, y >= 2^lod do not exist; upstream a bottom_right > 1 would invent them).
TileRange overlapping_tiles(const bt_preprocess_dataset& d, uint32_t lod) {
    const float n = float(1u << lod);
    auto sat = [&](float v) -> uint32_t { return !(v > 0.0f) ? 0u : (v >= n ? uint32_t(1u << lod) : uint32_t(v)); };
    return {sat(d.top_left[0] * n), sat(d.top_left[1] * n), sat(std::ceil(d.bottom_right[0] * n)), sat(std::ceil(d.bottom_right[1] * n))};
}

bt_status add_raster(bt_preprocessor* p, const bt_atlas* a, uint32_t ai, const bt_raster* src, int32_t* index) {
    if (!src || !src->data || !src->width || !src->height) {
        set_error("source raster missing");
        return BT_ERR_INVALID_ARGUMENT;
    }
    const uint32_t fmt = a->attachments[ai].meta.format;
    if (fmt != BT_FORMAT_R16 && fmt != BT_FORMAT_RGBA8) {
        set_error("attachment format %u is not processed (reference: preprocessing.wgsl:73-90 has no branch for it)", fmt);
        return BT_ERR_UNSUPPORTED;
    }
    if (src->format != fmt) {
        set_error("raster format %u != attachment format %u", src->format, fmt);
        return BT_ERR_INVALID_ARGUMENT;
    }
    const uint32_t px = fmt == BT_FORMAT_R16 ? 2 : 4;
    const uint64_t pitch = src->row_pitch ? src->row_pitch : uint64_t(src->width) * px;
    if (pitch < uint64_t(src->width) * px || pitch % px) {
        set_error("row_pitch %llu: must hold %u texels of %u bytes and be a multiple of the texel size", (unsigned long long)pitch, src->width, px);
        return BT_ERR_INVALID_ARGUMENT;
    }
    Raster r;
    r.format = fmt;
    r.owned = false;
    r.dev = {src->data, src->width, src->height, pitch};
    if (src->on_device > BT_RASTER_HOST_DEFERRED) {
        set_error("bt_raster.on_device = %u", src->on_device);
        return BT_ERR_INVALID_ARGUMENT;
    }
    // fused_main moves the source in 16-byte pieces (LDS-DMA, or 16-byte register loads): base and pitch must be multiples of 16 bytes, else
    // it stages texel by texel — a 16380-texel-wide R16 raster ran 3 x slower than the 16384-wide one (0.81 vs 0.27 ms).  So the library pads
    // what it uploads itself, and copies a BORROWED device raster whose base or pitch is not 16-byte aligned into a padded buffer of its own
    // (0.5 GB: 0.15 ms) when the queue first runs — the caller's memory is read then, as before.  Deferred host rasters keep the caller's pitch
    // (their bands travel as plain 1-D copies).
    const uint64_t row_bytes = uint64_t(src->width) * px, padded_pitch = (row_bytes + 15u) & ~uint64_t(15);
    const bool r16 = fmt == BT_FORMAT_R16;
    auto take_buffer = [&](uint64_t need, void** dev) -> bt_status {
        uint64_t kept_bytes = 0;
        if (void* kept = p->ctx->take_spare_raster(need, &kept_bytes)) {  // a buffer an earlier queue released
            *dev = kept;
            r.alloc_bytes = kept_bytes;
        } else {
            BT_HIP(hipMalloc(dev, need));
            r.alloc_bytes = need;
        }
        return BT_OK;
    };
    if (src->on_device == 1u && r16 && ((reinterpret_cast<uintptr_t>(src->data) | pitch) & 15u) != 0) {
        void* dev = nullptr;
        BT_HIP(hipSetDevice(p->ctx->device));
        if (bt_status st = take_buffer(padded_pitch * src->height, &dev)) return st;
        r.dev_src = src->data;  // copied (device to device, row by row) by the first run of the queue
        r.dev_src_pitch = pitch;
        r.pending = true;
        r.dev = {dev, src->width, src->height, padded_pitch};
        r.owned = true;
    } else if (src->on_device != 1u) {
        void* dev = nullptr;
        BT_HIP(hipSetDevice(p->ctx->device));
        // the caller's buffer ends with the last texel of the last row, not with a whole pitch
        const uint64_t bytes = pitch * (src->height - 1) + row_bytes;
        const bool pad = r16 && (pitch & 15u) != 0;  // (deferred rasters too: their windows / bands then travel as pitched copies)
        const uint64_t dev_pitch = pad ? padded_pitch : pitch;
        if (bt_status st = take_buffer(dev_pitch * src->height, &dev)) return st;
        if (src->on_device == BT_RASTER_HOST_DEFERRED) {  // copied when the queue runs; the caller keeps the rows alive until then
            r.host = src->data;
            r.host_bytes = bytes;
            r.host_pitch = pitch;
            r.pending = true;
        } else {
            hipError_t e = pad ? hipMemcpy2DAsync(dev, dev_pitch, src->data, pitch, row_bytes, src->height, hipMemcpyHostToDevice, p->ctx->stream)
                               : hipMemcpyAsync(dev, src->data, bytes, hipMemcpyHostToDevice, p->ctx->stream);
            if (e == hipSuccess) e = hipStreamSynchronize(p->ctx->stream);
            if (e != hipSuccess) {
                hipFree(dev);
                return hip_fail(e, "raster upload");
            }
        }
        r.dev.data = dev;
        r.dev.pitch = dev_pitch;
        r.owned = true;
    }
    *index = int32_t(p->rasters.size());
    p->rasters.push_back(r);
    return BT_OK;
}

Task make_task(TaskType type, const bt_atlas_tile& tile, const bt_preprocess_dataset& d, uint32_t job) {
    Task t{};
    t.type = type;
    t.coord = tile.coordinate;
    t.atlas_index = tile.atlas_index;
    t.attachment_index = d.attachment_index;
    t.tl[0] = d.top_left[0];
    t.tl[1] = d.top_left[1];
    t.br[0] = d.bottom_right[0];
    t.br[1] = d.bottom_right[1];
    t.raster = -1;
    t.job = job;
    return t;
}

void push_barrier(bt_preprocessor* p, uint32_t job) {
    Task t{};
    t.type = kBarrier;
    t.raster = -1;
    t.job = job;
    p->queue.push_back(t);
}

// Preprocessor::split_and_downsample (preprocessor.rs:234-269)
bt_status split_and_downsample(bt_preprocessor* p, bt_atlas* a, const bt_preprocess_dataset& d, int32_t raster, uint32_t job) {
    uint32_t lod = d.lod_end - 1;
    TileRange r = overlapping_tiles(d, lod);
    for (uint32_t x = r.lx; x < r.ux; x++)
        for (uint32_t y = r.ly; y < r.uy; y++) {
            bt_atlas_tile tile;
            if (bt_status s = bt_atlas_get_or_allocate_tile(a, {d.side, lod, x, y}, &tile)) return s;
            Task t = make_task(kSplit, tile, d, job);
            t.raster = raster;
            p->queue.push_back(t);
        }
    while (lod > d.lod_begin) {
        lod--;
        push_barrier(p, job);
        r = overlapping_tiles(d, lod);
        for (uint32_t x = r.lx; x < r.ux; x++)
            for (uint32_t y = r.ly; y < r.uy; y++) {
                bt_atlas_tile tile;
                if (bt_status s = bt_atlas_get_or_allocate_tile(a, {d.side, lod, x, y}, &tile)) return s;
                Task t = make_task(kDownsample, tile, d, job);
                bt_tile_coordinate ch[4];
                tile_children(tile.coordinate, ch);
                for (int i = 0; i < 4; i++) bt_atlas_get_tile(a, ch[i], &t.rel[i]);
                p->queue.push_back(t);
            }
    }
    return BT_OK;
}

// Preprocessor::stitch_and_save_layer (preprocessor.rs:271-288)
bt_status stitch_and_save_layer(bt_preprocessor* p, bt_atlas* a, const bt_preprocess_dataset& d, uint32_t lod, uint32_t job) {
    const TileRange r = overlapping_tiles(d, lod);
    for (int pass = 0; pass < 2; pass++) {
        for (uint32_t x = r.lx; x < r.ux; x++)
            for (uint32_t y = r.ly; y < r.uy; y++) {
                bt_atlas_tile tile;
                if (bt_status s = bt_atlas_get_or_allocate_tile(a, {d.side, lod, x, y}, &tile)) return s;
                Task t = make_task(pass == 0 ? kStitch : kSave, tile, d, job);
                if (pass == 0) {
                    bt_tile_coordinate nb[8];
                    tile_neighbours(tile.coordinate, a->config.spherical != 0, nb);
                    for (int i = 0; i < 8; i++) bt_atlas_get_tile(a, nb[i], &t.rel[i]);
                }
                p->queue.push_back(t);
            }
        if (pass == 0) push_barrier(p, job);
    }
    return BT_OK;
}

bt_status check_dataset(const bt_atlas* a, uint32_t ai, uint32_t lod_begin, uint32_t lod_end) {
    if (ai >= a->attachments.size()) {
        set_error("attachment_index %u out of range", ai);
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (lod_end <= lod_begin || lod_end > 31) {
        set_error("lod_range %u..%u", lod_begin, lod_end);
        return BT_ERR_INVALID_ARGUMENT;
    }
    const AttachmentMeta& m = a->attachments[ai].meta;
    if (lod_end - lod_begin > 1 && (m.center_size & 1u)) {
        // downsample.wgsl:18-20 indexes child_tiles[0..4) with center_size/2; odd sizes run out of bounds upstream
        set_error("center_size %u must be even to downsample", m.center_size);
        return BT_ERR_UNSUPPORTED;
    }
    if (m.center_size < m.border_size) {
        // stitch.wgsl:79-88 then reads a neighbour's apron (q = p +- c lands outside its centre) while that apron is being
        // written by the same pass: undefined upstream, refused here
        set_error("center_size %u < border_size %u", m.center_size, m.border_size);
        return BT_ERR_UNSUPPORTED;
    }
    if ((uint64_t(m.texture_size) * m.pixel_size) % 4u) {
        set_error("texture row must be a whole number of 32-bit entries");
        return BT_ERR_UNSUPPORTED;
    }
    return BT_OK;
}

}  // namespace

extern "C" {

// Preprocessor::preprocess_tile (preprocessor.rs:298-312)
bt_status bt_preprocessor_preprocess_tile(bt_preprocessor* p, bt_atlas* a, const bt_preprocess_dataset* d, const bt_raster* src) {
    if (!p || !a || !d) return BT_ERR_INVALID_ARGUMENT;
    if (bt_status s = check_dataset(a, d->attachment_index, d->lod_begin, d->lod_end)) return s;
    if (d->side >= 6u) {  // a cube has six sides (coordinate.rs:17-40: the side tables); a planar terrain uses side 0
        set_error("dataset side %u", d->side);
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (!(d->bottom_right[0] > d->top_left[0]) || !(d->bottom_right[1] > d->top_left[1])) {
        set_error("empty dataset rectangle");
        return BT_ERR_INVALID_ARGUMENT;
    }
    int32_t raster;
    if (bt_status s = add_raster(p, a, d->attachment_index, src, &raster)) return s;
    const uint32_t job = p->jobs++;
    p->compiled = false;
    p->saves_recorded = false;
    if (bt_status s = split_and_downsample(p, a, *d, raster, job)) return s;
    push_barrier(p, job);
    for (uint32_t lod = d->lod_begin; lod < d->lod_end; lod++)
        if (bt_status s = stitch_and_save_layer(p, a, *d, lod, job)) return s;
    return BT_OK;
}

// Preprocessor::preprocess_spherical (preprocessor.rs:314-343)
bt_status bt_preprocessor_preprocess_spherical(bt_preprocessor* p, bt_atlas* a, const bt_spherical_dataset* sd, const bt_raster sources[6]) {
    if (!p || !a || !sd || !sources) return BT_ERR_INVALID_ARGUMENT;
    if (bt_status s = check_dataset(a, sd->attachment_index, sd->lod_begin, sd->lod_end)) return s;
    const uint32_t job = p->jobs++;
    p->compiled = false;
    p->saves_recorded = false;
    bt_preprocess_dataset side[6];
    int32_t raster[6];
    for (uint32_t s = 0; s < 6; s++) {
        side[s] = {sd->attachment_index, s, {0.0f, 0.0f}, {1.0f, 1.0f}, sd->lod_begin, sd->lod_end};
        if (bt_status rc = add_raster(p, a, sd->attachment_index, &sources[s], &raster[s])) return rc;
    }
    for (uint32_t s = 0; s < 6; s++)
        if (bt_status rc = split_and_downsample(p, a, side[s], raster[s], job)) return rc;
    push_barrier(p, job);
    for (uint32_t lod = sd->lod_begin; lod < sd->lod_end; lod++)
        for (uint32_t s = 0; s < 6; s++)
            if (bt_status rc = stitch_and_save_layer(p, a, side[s], lod, job)) return rc;
    return BT_OK;
}

uint32_t bt_preprocessor_task_counts(const bt_preprocessor* p, uint32_t counts[5]) {
    if (!p) return 0;
    if (counts) {
        memset(counts, 0, 5 * sizeof(uint32_t));
        for (const Task& t : p->queue) counts[t.type]++;
    }
    return uint32_t(p->queue.size());
}

bt_status bt_preprocessor_save(bt_preprocessor* p, bt_atlas* a, const char* assets_root) {
    if (!p || !a || !assets_root) return BT_ERR_INVALID_ARGUMENT;
    // exactly the tiles of the Save tasks that have run (select_ready_tasks -> tile_atlas.save, preprocessor.rs:378-380)
    const std::string terrain = std::string(assets_root) + "/" + a->config.path;
    for (uint32_t ai = 0; ai < a->attachments.size(); ai++) {
        std::vector<std::pair<uint32_t, bt_tile_coordinate>> tiles;
        for (const AtlasTileAttachment& t : a->to_save) {
            if (t.attachment_index != ai || t.atlas_index == BT_INVALID_ATLAS_INDEX) continue;
            // after a distributed sharded run a rank writes its share only: the finest tiles it computed, and of the lower
            // LODs (complete on every rank) every world-th tile — each file has exactly one writer
            if (p->shard_world > 1 && p->shard_distributed && shard_holder(p, ai, t.coordinate.lod, t.atlas_index) != p->shard_rank) continue;
            tiles.push_back({t.atlas_index, t.coordinate});
        }
        if (tiles.empty()) continue;
        // AtlasAttachment::new: path = "assets/{path}/data/{name}" (tile_atlas.rs:175)
        const std::string dir = terrain + "/data/" + a->attachments[ai].cfg.name;
        if (bt_status s = save_tiles(a, ai, dir.c_str(), std::move(tiles))) return s;
    }
    a->to_save.clear();
    p->saves_recorded = false;
    if (p->shard_world > 1 && p->shard_distributed && p->shard_rank != 0) return BT_OK;  // config.tc: rank 0
    if (bt_status s = make_dirs(terrain)) return s;
    return bt_atlas_save_tile_config(a, (terrain + "/config.tc").c_str());
}

// ------------------------------------------------------------------------------------------------ streamed run
}  // extern "C"

namespace bt {
bool fused_source_window(const bt_preprocessor* p, uint32_t raster, uint32_t out[4]);
bt_status shard_exchange(bt_preprocessor* p, bt_atlas* a, bt_comm* comm, hipStream_t stream, bool distributed);  // bt_comm.cpp: the grouped collective of a sharded step
bt_status shard_check_comm(const bt_preprocessor* p, const bt_comm* comm);

// Deferred host rasters travel when the queue runs.  A SHARDED preprocessor (compiled plan known) uploads only the texels its
// own launches read — its column strips + halo (a rank never touches the rest of the source).  What has travelled is
// remembered per raster (Raster::windows, every rectangle the device holds): when a kept queue is compiled again — another rank / world
// (bt_preprocessor_set_shard), BT_RUN_GENERIC or BT_RUN_REFERENCE_DISPATCH, whose launches read the whole raster — and its launches read
// texels outside every such rectangle, the missing window travels before the run (the caller keeps the rows of a deferred raster alive
// until the queue is RELEASED: the ABI-6 lifetime rule of bt_raster).  A borrowed device raster that is not 16-byte aligned is copied into
// its padded buffer by EVERY run ("borrowed" means "read at run time", whatever the width).  skip[i] != 0: raster i is handled by the
// caller (the streamed run uploads it band by band).
bt_status upload_pending_rasters(bt_preprocessor* p, const std::vector<uint8_t>* skip) {
    for (size_t i = 0; i < p->rasters.size(); i++) {
        Raster& r = p->rasters[i];
        if (skip && i < skip->size() && (*skip)[i]) continue;
        if (r.dev_src) {
            const uint64_t px2 = r.format == BT_FORMAT_R16 ? 2 : 4;
            BT_HIP(hipMemcpy2DAsync((void*)r.dev.data, r.dev.pitch, r.dev_src, r.dev_src_pitch, uint64_t(r.dev.width) * px2, r.dev.height, hipMemcpyDeviceToDevice, p->ctx->stream));
            r.pending = false;
            continue;
        }
        if (!r.host) continue;  // not a deferred raster
        uint32_t w[4] = {0u, 0u, r.dev.width, r.dev.height};
        const uint64_t px = r.format == BT_FORMAT_R16 ? 2 : 4;
        const bool window = p->shard_world > 1 && p->compiled && fused_source_window(p, uint32_t(i), w);
        if (!window) {
            w[0] = w[1] = 0;
            w[2] = r.dev.width;
            w[3] = r.dev.height;
        }
        const bool empty = !(w[2] > w[0] && w[3] > w[1]);
        const bool covered = empty || r.holds(w);
        if (!r.pending && covered) continue;
        p->uploaded_source_bytes = 0;  // (the last deferred raster that was looked at: an empty window travels as 0 bytes)
        if (covered) {
            r.pending = false;
            continue;
        }
        if (w[0] == 0 && w[1] == 0 && w[2] == r.dev.width && w[3] == r.dev.height && r.host_pitch == r.dev.pitch) {
            BT_HIP(hipMemcpyAsync((void*)r.dev.data, r.host, r.host_bytes, hipMemcpyHostToDevice, p->ctx->stream));
            p->uploaded_source_bytes = r.host_bytes;
        } else {  // a window, or a padded device copy: pitched
            const uint64_t off_dev = uint64_t(w[1]) * r.dev.pitch + uint64_t(w[0]) * px, off_host = uint64_t(w[1]) * r.host_pitch + uint64_t(w[0]) * px;
            BT_HIP(hipMemcpy2DAsync((uint8_t*)r.dev.data + off_dev, r.dev.pitch, (const uint8_t*)r.host + off_host, r.host_pitch, (w[2] - w[0]) * px, w[3] - w[1],
                                    hipMemcpyHostToDevice, p->ctx->stream));
            p->uploaded_source_bytes = uint64_t(w[2] - w[0]) * px * (w[3] - w[1]);
        }
        BT_HIP(hipStreamSynchronize(p->ctx->stream));
        r.add_window(w);
        r.pending = false;
    }
    return BT_OK;
}
}  // namespace bt

namespace {
// The upload and download queues of the streamed run.  They get NON-DEFAULT PRIORITIES — not for the priority's sake: the runtime maps HIP
// streams onto a handful of hardware queues round-robin PER PRIORITY CLASS, and a process that owns a few other default-priority streams
// (a host application does; bench.py's second lane does) can land the download stream on the kernels' own hardware queue, where every copy
// then waits behind the next bands' kernels: config 2 end to end 6.6 -> 8.9 ms with exactly one extra stream in the process (round 6
// probe).  A class of their own keeps the three queues apart whatever else the process has created.
bt_status ctx_side_streams(bt_ctx* ctx) {
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) least = greatest = 0;
    if (!ctx->copy_stream) BT_HIP(hipStreamCreateWithPriority(&ctx->copy_stream, hipStreamNonBlocking, greatest));
    if (!ctx->save_stream) BT_HIP(hipStreamCreateWithPriority(&ctx->save_stream, hipStreamNonBlocking, least));
    return BT_OK;
}

// One step of a streamed run: an upload (optional), a launch — a whole plan entry or a band of a fused main / direct launch — and the
// tiles that are complete, and may leave, once that launch has run.
struct StreamStep {
    size_t plan_index = 0;
    bool band = false;
    uint32_t item_begin = 0, item_count = 0;
    int32_t raster = -1;        // band: rows below `row_end` of this deferred raster travel first (those that have not yet)
    uint32_t row_end = 0;
    uint32_t attachment = 0;
    TileSaver::Tiles early;     // the band's finished finest tiles
    TileSaver::Tiles rest;      // behind the attachment's last launch: every tile of it that has not left yet
    bool exchange_before = false;  // a sharded run with a communicator: the grouped collective precedes this step's launch
};

bt_status run_streamed_impl(bt_preprocessor* p, bt_atlas* a, bt_comm* comm, const char* assets_root, uint32_t flags, bt_stream_stats* out) {
    if (!p || !a || !assets_root) return BT_ERR_INVALID_ARGUMENT;
    if (p->ctx != a->ctx) {
        set_error("preprocessor and atlas belong to different contexts");
        return BT_ERR_INVALID_ARGUMENT;
    }
    const bool sharded = p->shard_world > 1;
    uint32_t halves = flags & (BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_FINISH);
    if (!sharded || !halves) halves = BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_FINISH;
    const bool local = (halves & BT_RUN_SHARD_LOCAL) != 0, finish = (halves & BT_RUN_SHARD_FINISH) != 0;
    if (sharded && local && finish && !comm) {
        set_error("bt_preprocessor_run_streamed_sharded: both halves in one call need a communicator (or call BT_RUN_SHARD_LOCAL, exchange, BT_RUN_SHARD_FINISH)");
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(p->ctx->device));
    bt_stream_stats st{};
    const uint32_t mode = flags & (BT_RUN_GENERIC | BT_RUN_REFERENCE_DISPATCH);
    if (bt_status s = ensure_compiled(p, a, mode)) return s;
    if (sharded) {
        // only the distributed result makes sense here (a replicated atlas has no "share" to write): the finest LOD stays where it was computed
        if (p->shard_pieces.empty()) {
            set_error("bt_preprocessor_run_streamed_sharded: the queue does not shard (world %u must divide its units; fused plans only)", p->shard_world);
            return BT_ERR_UNSUPPORTED;
        }
        for (const bt_shard_piece& piece : p->shard_pieces)
            if (piece.side != p->shard_pieces[0].side) {
                set_error("BT_RUN_SHARD_DISTRIBUTED needs a one-sided (planar) job: cube seams read finest tiles of other ranks");
                return BT_ERR_UNSUPPORTED;
            }
        p->shard_distributed = true;
    }
    if (!p->saves_recorded) {
        for (const Task& t : p->queue)
            if (t.type == kSave) a->to_save.push_back({t.coord, t.atlas_index, t.attachment_index});
        p->saves_recorded = true;
    }

    // ---- the steps: the plan's entries in launch order (a sharded step: the local half, the exchange, the finishing half), bandable
    // launches cut into bands
    uint32_t rows_per_band = 0;  // automatic
#ifdef BT_DEBUG_HOOKS
    if (const char* e = getenv("BT_STREAM_BAND_ROWS")) rows_per_band = uint32_t(std::max(1, atoi(e)));
#endif
    std::vector<StreamStep> steps;
    std::vector<uint8_t> banded_raster(p->rasters.size(), 0);
    std::vector<std::vector<StreamBand>> bands_of(p->plan.size());
    for (int half = 0; half < 2; half++) {
        if (half == 0 ? !local : !finish) continue;
        bool first_of_half = true;
        for (size_t i = 0; i < p->plan.size(); i++) {
            const Launch& l = p->plan[i];
            if (sharded ? (l.phase == 2) != (half == 1) : half == 1) continue;
            std::vector<StreamBand>& bands = bands_of[i];
            bool bandable = fused_stream_bands(p, l, rows_per_band, &bands) && !bands.empty();
            for (const StreamBand& b : bands) {
                const Raster& r = p->rasters[b.raster];
                bandable = bandable && r.host != nullptr && r.pending && !r.dev_src;  // a deferred host raster that has not travelled
            }
            StreamStep proto;
            proto.plan_index = i;
            proto.attachment = l.attachment;
            proto.exchange_before = sharded && comm && half == 1 && first_of_half && local;
            first_of_half = false;
            if (!bandable) {
                bands.clear();
                steps.push_back(proto);
                continue;
            }
            for (size_t k = 0; k < bands.size(); k++) {
                StreamStep sb = proto;
                sb.exchange_before = proto.exchange_before && k == 0;
                sb.band = true;
                sb.item_begin = bands[k].item_begin;
                sb.item_count = bands[k].item_count;
                sb.raster = int32_t(bands[k].raster);
                // (the last band of a raster takes the rest of it: rows below the last tile row's apron that no kernel reads still count as uploaded)
                const bool last_of_raster = k + 1 == bands.size() || bands[k + 1].raster != bands[k].raster;
                sb.row_end = last_of_raster ? p->rasters[bands[k].raster].dev.height : bands[k].source_row_end;
                banded_raster[bands[k].raster] = 1;
                steps.push_back(sb);
            }
            st.banded_launches++;
            st.bands += uint32_t(bands.size());
        }
    }
    // (a sharded one-call run whose plan has no finishing launch — a single-LOD job — still owes the step its collective)
    bool exchange_scheduled = false;
    for (const StreamStep& sp : steps) exchange_scheduled = exchange_scheduled || sp.exchange_before;
    const bool exchange_at_end = sharded && comm && local && finish && !exchange_scheduled;
    const bool streamable = st.bands > 1;
    if (!streamable) {  // the same result, one leg after the other
        const uint32_t keep = mode | BT_RUN_KEEP_QUEUE;
        if (!sharded) {
            if (bt_status s = bt_preprocessor_run(p, a, keep)) return s;
        } else {
            if (local)
                if (bt_status s = bt_preprocessor_run(p, a, keep | BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_DISTRIBUTED)) return s;
            if (local && finish)
                if (bt_status s = shard_exchange(p, a, comm, p->ctx->stream, true)) return s;
            if (finish)
                if (bt_status s = bt_preprocessor_run(p, a, keep | BT_RUN_SHARD_FINISH | BT_RUN_SHARD_DISTRIBUTED)) return s;
        }
        bt_stream_stats none{};
        if (finish) {
            for (const AtlasTileAttachment& t : a->to_save)  // what bt_preprocessor_save is about to write (a sharded rank: its share)
                if (t.atlas_index != BT_INVALID_ATLAS_INDEX && (!sharded || shard_holder(p, t.attachment_index, t.coordinate.lod, t.atlas_index) == p->shard_rank))
                    none.saved_bytes += a->attachments[t.attachment_index].tile_bytes;
            if (bt_status s = bt_preprocessor_save(p, a, assets_root)) return s;
        }
        if (out) *out = none;
        return ((flags & BT_RUN_KEEP_QUEUE) || !finish) ? BT_OK : release_queue(p);
    }
    if (bt_status s = ctx_side_streams(p->ctx)) return s;

    // ---- which tiles leave after which step.  A finest tile of a banded launch leaves with its band when nothing later writes it: the
    // attachment has ONE job in the queue (an overlay or an adjacent dataset would write or stitch it again), and on a cube it does not
    // touch a face edge (its cross-face aprons are stitched after the last face).  Everything else of an attachment leaves behind the
    // attachment's last launch.  A sharded rank writes its share only (shard_holder).
    const std::string terrain = std::string(assets_root) + "/" + a->config.path;
    auto dir_of = [&](uint32_t ai) { return terrain + "/data/" + a->attachments[ai].cfg.name; };
    std::vector<uint32_t> jobs_of(a->attachments.size(), 0);
    {
        std::vector<std::vector<uint32_t>> seen(a->attachments.size());
        for (const Task& t : p->queue)
            if (t.type == kSplit && std::find(seen[t.attachment_index].begin(), seen[t.attachment_index].end(), t.job) == seen[t.attachment_index].end()) {
                seen[t.attachment_index].push_back(t.job);
                jobs_of[t.attachment_index]++;
            }
    }
    std::vector<uint8_t> in_plan(a->attachments.size(), 0);
    for (const StreamStep& sp : steps) in_plan[sp.attachment] = 1;
    // to_save entries of the attachments this run handles, by (attachment, atlas index)
    std::vector<std::unordered_map<uint32_t, bt_tile_coordinate>> waiting(a->attachments.size());
    for (const AtlasTileAttachment& t : a->to_save) {
        if (t.atlas_index == BT_INVALID_ATLAS_INDEX || !in_plan[t.attachment_index]) continue;
        if (sharded && shard_holder(p, t.attachment_index, t.coordinate.lod, t.atlas_index) != p->shard_rank) continue;
        waiting[t.attachment_index][t.atlas_index] = t.coordinate;
    }
    const bool spherical = a->config.spherical != 0;
    for (StreamStep& sp : steps) {
        if (!sp.band || jobs_of[sp.attachment] != 1) continue;
        std::vector<FusedTile> tiles;
        fused_launch_tiles(p, p->plan[sp.plan_index], sp.item_begin, sp.item_count, &tiles);
        for (const FusedTile& t : tiles) {
            const bt_tile_coordinate& c = t.coordinate;
            const uint32_t n = 1u << c.lod;
            if (spherical && (c.x == 0 || c.y == 0 || c.x == n - 1 || c.y == n - 1)) continue;
            auto w = waiting[sp.attachment].find(t.atlas_index);
            if (w == waiting[sp.attachment].end() || !operator_eq(w->second, c)) continue;
            sp.early.push_back({t.atlas_index, c});
            waiting[sp.attachment].erase(w);
        }
        st.early_tiles += uint32_t(sp.early.size());
    }
    if (finish)
        for (uint32_t ai = 0; ai < a->attachments.size(); ai++) {
            if (!in_plan[ai] || waiting[ai].empty()) continue;
            size_t last = steps.size();
            for (size_t k = 0; k < steps.size(); k++)
                if (steps[k].attachment == ai) last = k;
            for (const auto& [index, coord] : waiting[ai]) steps[last].rest.push_back({index, coord});
        }
    size_t last_saving = 0;
    for (size_t k = 0; k < steps.size(); k++)
        if (!steps[k].early.empty() || !steps[k].rest.empty()) last_saving = k;

    const size_t ns = steps.size();
    std::vector<hipEvent_t> computed(ns, nullptr);
    hipEvent_t uploaded = nullptr;
    bt_status rc = BT_OK;
    for (hipEvent_t& e : computed)
        if (rc == BT_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) rc = BT_ERR_DEVICE;
    if (rc == BT_OK && hipEventCreateWithFlags(&uploaded, hipEventDisableTiming) != hipSuccess) rc = BT_ERR_DEVICE;

#ifdef BT_DEBUG_HOOKS
    g_trace = getenv("BT_STREAM_TRACE") != nullptr;
    g_trace_start = std::chrono::steady_clock::now();
#endif
    auto stamp = [](const char* what, size_t k) { trace_stamp(what, k); };
    // the saver: step after step as their kernels are enqueued (host handshake), ordered on the GPU by events
    std::mutex m;
    std::condition_variable cv;
    size_t launched = 0;  // steps whose `computed` event has been recorded
    bool abort_run = false;
    bt_status save_rc = BT_OK;
    char save_error[512] = "";
    uint64_t saved_bytes = 0;
    std::thread saver([&] {
        hipSetDevice(p->ctx->device);
        TileSaver ts(a, p->ctx->save_stream);
        bt_status s = ts.begin();
        for (size_t k = 0; k < ns && s == BT_OK;) {
            if (steps[k].early.empty() && steps[k].rest.empty()) {
                k++;
                continue;
            }
            // The saver takes what is ready: step k and every following step of the same attachment that the launcher has enqueued by
            // now travel as ONE hand-over (sorted by layer, cut into 32 MB chunks).  When the download + write side is the slower one — it
            // is, on PCIe — the bands it falls behind on merge into full-size chunks instead of paying the per-chunk latencies band by band
            // (config 2's 8 MB and 16 MB bands: 0.6 / 0.9 ms each, i.e. 13 - 17 GB/s; merged 32 MB chunks move at 45).
            size_t last = k;
            {
                std::unique_lock<std::mutex> lock(m);
                cv.wait(lock, [&] { return launched > k || abort_run; });
                if (abort_run) break;
                while (last + 1 < ns && launched > last + 1 && steps[last + 1].attachment == steps[k].attachment) last++;
            }
            if (hipStreamWaitEvent(p->ctx->save_stream, computed[last], 0) != hipSuccess) s = BT_ERR_DEVICE;
            stamp("saver: steps taken up to", last);
            TileSaver::Tiles tiles;
            for (size_t q = k; q <= last; q++) {
                tiles.insert(tiles.end(), steps[q].early.begin(), steps[q].early.end());
                tiles.insert(tiles.end(), steps[q].rest.begin(), steps[q].rest.end());
            }
            if (s == BT_OK && !tiles.empty()) s = ts.add(steps[k].attachment, dir_of(steps[k].attachment), std::move(tiles), last >= last_saving);
            stamp("saver: copies issued, previous chunks handed to the writers", last);
            k = last + 1;
        }
        if (s == BT_OK) s = ts.finish();
        saved_bytes = ts.saved_bytes();
        if (s != BT_OK) {
            snprintf(save_error, sizeof save_error, "%s", bt_last_error());
            save_rc = s;
        }
    });

    // this thread: upload what a step needs (a pageable copy holds the host until it is done; the GPU meanwhile runs the step before), launch it
    auto publish = [&](size_t n) {
        { std::lock_guard<std::mutex> lock(m); launched = n; }
        cv.notify_all();
    };
    // rasters no band covers (a launch that cannot be banded reads them): whole, up front, on the kernels' stream
    if (rc == BT_OK && local) rc = upload_pending_rasters(p, &banded_raster);
    if (rc == BT_OK && local) p->stats.prev_zero_launches = fused_begin_run(p, a);
    // per banded raster: the column window that travels (a sharded rank: its strips + halo) and the rows that have
    std::vector<std::array<uint32_t, 4>> window(p->rasters.size());
    std::vector<uint32_t> done_rows(p->rasters.size(), 0);
    for (size_t i = 0; i < p->rasters.size() && rc == BT_OK; i++) {
        if (!banded_raster[i]) continue;
        const Raster& r = p->rasters[i];
        uint32_t w[4] = {0u, 0u, r.dev.width, r.dev.height};
        if (!(sharded && fused_source_window(p, uint32_t(i), w))) {
            w[0] = w[1] = 0;
            w[2] = r.dev.width;
            w[3] = r.dev.height;
        }
        window[i] = {w[0], w[1], w[2], w[3]};
        done_rows[i] = w[1];
    }
    for (size_t k = 0; k < ns && rc == BT_OK; k++) {
        const StreamStep& sp = steps[k];
        const Launch& l = p->plan[sp.plan_index];
        if (sp.exchange_before) rc = shard_exchange(p, a, comm, p->ctx->stream, true);
        if (rc != BT_OK) break;
        if (sp.band) {
            Raster& r = p->rasters[size_t(sp.raster)];
            const std::array<uint32_t, 4>& w = window[size_t(sp.raster)];
            const uint32_t end_row = std::min(sp.row_end, w[3]);
            uint32_t& done = done_rows[size_t(sp.raster)];
            if (end_row > done && w[2] > w[0]) {
                stamp("upload begin", k);
                const uint64_t px = r.format == BT_FORMAT_R16 ? 2 : 4;
                const uint8_t* host = (const uint8_t*)r.host;
                uint8_t* dev = (uint8_t*)r.dev.data;
                hipError_t ce;
                if (r.host_pitch == r.dev.pitch && w[0] == 0 && w[2] == r.dev.width) {
                    const uint64_t off = uint64_t(done) * r.dev.pitch, end = std::min<uint64_t>(r.host_bytes, uint64_t(end_row) * r.dev.pitch);
                    ce = hipMemcpyAsync(dev + off, host + off, end - off, hipMemcpyHostToDevice, p->ctx->copy_stream);
                    st.uploaded_bytes += end - off;
                } else {  // a column window (a sharded rank's strips) or a padded device copy (the caller's rows are not 16-byte aligned): pitched
                    ce = hipMemcpy2DAsync(dev + uint64_t(done) * r.dev.pitch + w[0] * px, r.dev.pitch, host + uint64_t(done) * r.host_pitch + w[0] * px, r.host_pitch,
                                          uint64_t(w[2] - w[0]) * px, end_row - done, hipMemcpyHostToDevice, p->ctx->copy_stream);
                    st.uploaded_bytes += uint64_t(w[2] - w[0]) * px * (end_row - done);
                }
                if (ce != hipSuccess) rc = BT_ERR_DEVICE;
                stamp("upload call returned", k);
                done = end_row;
                if (rc == BT_OK && hipEventRecord(uploaded, p->ctx->copy_stream) != hipSuccess) rc = BT_ERR_DEVICE;
                if (rc == BT_OK && hipStreamWaitEvent(p->ctx->stream, uploaded, 0) != hipSuccess) rc = BT_ERR_DEVICE;
            }
            if (rc == BT_OK) rc = fused_launch_range(p, a, l, sp.item_begin, sp.item_count);
        } else {
            rc = run_plan_entry(p, a, l);
        }
        if (rc == BT_OK && hipEventRecord(computed[k], p->ctx->stream) != hipSuccess) rc = BT_ERR_DEVICE;
        if (rc == BT_OK) publish(k + 1);
    }
    if (rc == BT_OK && exchange_at_end) rc = shard_exchange(p, a, comm, p->ctx->stream, true);
    // a raster counts as uploaded only when every band of it went out; after a failure a later run of the kept queue uploads it whole
    if (rc == BT_OK)
        for (size_t i = 0; i < p->rasters.size(); i++)
            if (banded_raster[i]) {
                Raster& r = p->rasters[i];
                const uint32_t w[4] = {window[i][0], window[i][1], window[i][2], window[i][3]};
                r.pending = false;
                if (w[2] > w[0] && w[3] > w[1]) r.add_window(w);
            }
    p->uploaded_source_bytes = st.uploaded_bytes;
    if (rc != BT_OK) {
        { std::lock_guard<std::mutex> lock(m); abort_run = true; }
        cv.notify_all();
    }
    stamp("all launched", ns);
    saver.join();
    stamp("saver done", ns);
    hipStreamSynchronize(p->ctx->stream);
    if (rc != BT_OK || save_rc != BT_OK) {  // nothing of this call may still read the caller's raster or write the pinned buffers
        hipStreamSynchronize(p->ctx->copy_stream);
        hipStreamSynchronize(p->ctx->save_stream);
    }
    for (hipEvent_t e : computed) if (e) hipEventDestroy(e);
    if (uploaded) hipEventDestroy(uploaded);
    if (rc == BT_ERR_DEVICE) set_error("bt_preprocessor_run_streamed: HIP call failed (%s)", hipGetErrorString(hipGetLastError()));
    if (rc != BT_OK) return rc;
    if (save_rc != BT_OK) {
        set_error("%s", save_error);
        return save_rc;
    }
    st.saved_bytes = saved_bytes;
    // What the saver wrote leaves the atlas's list.  A finishing call wrote every entry of the plan's attachments (a sharded rank: its
    // share — the others' entries go too, their holders write them); a local-only call only the early tiles.  Whatever else waits (Save
    // tasks of another attachment from an earlier run that was not saved yet) goes through bt_preprocessor_save, which also writes config.tc.
    a->to_save.erase(std::remove_if(a->to_save.begin(), a->to_save.end(),
                                    [&](const AtlasTileAttachment& t) {
                                        if (!in_plan[t.attachment_index]) return false;
                                        if (finish) return true;
                                        const auto& w = waiting[t.attachment_index];
                                        const bool mine = !sharded || shard_holder(p, t.attachment_index, t.coordinate.lod, t.atlas_index) == p->shard_rank;
                                        return mine && w.find(t.atlas_index) == w.end();  // (held by this rank and no longer waiting: it left with a band)
                                    }),
                     a->to_save.end());
    if (finish)
        if (bt_status s = bt_preprocessor_save(p, a, assets_root)) return s;
    st.streamed = 1;
    if (out) *out = st;
    return ((flags & BT_RUN_KEEP_QUEUE) || !finish) ? BT_OK : release_queue(p);
}
}  // namespace

extern "C" {

// The reference's own span (preprocessor.rs:363,419: sources loaded -> all saves done) as ONE overlapped pipeline: the source
// rasters travel to the GPU in bands of tile rows on a copy queue, each band's kernels start when its rows (and the few
// apron rows below it) have landed, and a second thread downloads and writes a band's finished tiles on a third queue while
// the next bands upload and run: H2D, kernels, D2H and the file system work at the same time (PCIe is full duplex).  Round 6: every
// fused main / direct launch of the plan is banded — several attachments (examples/preprocess_planar.rs:16-60), the six faces of a
// cube job (examples/preprocess_spherical.rs:20-48) — and a sharded rank streams its own window and share.
bt_status bt_preprocessor_run_streamed(bt_preprocessor* p, bt_atlas* a, const char* assets_root, uint32_t flags, bt_stream_stats* out) {
    if (p && p->shard_world > 1) {
        set_error("bt_preprocessor_run_streamed: a sharded preprocessor runs through bt_preprocessor_run_streamed_sharded");
        return BT_ERR_UNSUPPORTED;
    }
    return run_streamed_impl(p, a, nullptr, assets_root, flags, out);
}

bt_status bt_preprocessor_run_streamed_sharded(bt_preprocessor* p, bt_atlas* a, bt_comm* comm, const char* assets_root, uint32_t flags, bt_stream_stats* out) {
    if (!p) return BT_ERR_INVALID_ARGUMENT;
    if (comm && p->shard_world > 1)
        if (bt_status s = shard_check_comm(p, comm)) return s;
    return run_streamed_impl(p, a, comm, assets_root, flags, out);
}

bt_status bt_preprocessor_last_run_stats(const bt_preprocessor* p, bt_run_stats* out) {
    if (!p || !out) return BT_ERR_INVALID_ARGUMENT;
    *out = p->stats;
    return BT_OK;
}

}  // extern "C"
