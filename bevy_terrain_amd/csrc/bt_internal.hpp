// Internal declarations shared by the host (.cpp) and device (.hip) halves of libbevy_terrain_amd.so.
// Nothing here is part of the ABI; the ABI is include/bevy_terrain_amd.h.
#pragma once

#include <hip/hip_runtime.h>

#include <algorithm>
#include <array>
#include <cstdint>
#include <deque>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "bevy_terrain_amd.h"

struct bt_preprocessor;

namespace bt {

void set_error(const char* fmt, ...);
bt_status hip_fail(hipError_t e, const char* what);

#define BT_HIP(expr)                                          \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return bt::hip_fail(_e, #expr); \
    } while (0)

inline bool operator_eq(const bt_tile_coordinate& a, const bt_tile_coordinate& b) {
    return a.side == b.side && a.lod == b.lod && a.x == b.x && a.y == b.y;
}
inline bool is_invalid(const bt_tile_coordinate& c) {
    return c.side == 0xFFFFFFFFu && c.lod == 0xFFFFFFFFu && c.x == 0xFFFFFFFFu && c.y == 0xFFFFFFFFu;
}

struct CoordHash {
    size_t operator()(const bt_tile_coordinate& c) const {
        uint64_t h = (uint64_t(c.side) << 59) ^ (uint64_t(c.lod) << 53) ^ (uint64_t(c.x) << 26) ^ uint64_t(c.y);
        h ^= h >> 31;
        h *= 0x9E3779B97F4A7C15ull;
        return size_t(h ^ (h >> 29));
    }
};
struct CoordEq {
    bool operator()(const bt_tile_coordinate& a, const bt_tile_coordinate& b) const { return operator_eq(a, b); }
};

// ---- device-visible descriptors -------------------------------------------------------------

// AttachmentMeta of the reference (gpu_tile_atlas.rs:45-56) reduced to what the kernels read.
struct AttachmentMeta {
    uint32_t format;        // BT_FORMAT_R16 / BT_FORMAT_RGBA8
    uint32_t texture_size;  // T
    uint32_t border_size;   // b
    uint32_t center_size;   // c = T - 2b
    uint32_t atlas_size;    // layers
    uint32_t pixel_size;    // bytes
    uint32_t row_limit;     // rows [0, row_limit) of a tile are written by the batched kernels: T, or (T / 8) * 8 under BT_RUN_REFERENCE_DISPATCH
};

struct RasterDev {
    const void* data;
    uint32_t width, height;
    uint64_t pitch;  // bytes per row
};

// One queued Split / Downsample / Stitch task in the form the batched kernels read.
struct TaskDev {
    uint32_t atlas_index;
    uint32_t side, lod, x, y;
    float tlx, tly, brx, bry;
    uint32_t raster;
    uint32_t regions;       // stitch: bit r set = only apron region r (0 top .. 7 bottom-left) is written; 0 = all eight
    uint32_t rel_index[8];  // children (4) or neighbours (8): atlas indices, INVALID if absent
    uint32_t rel_side[8];   // neighbour sides (stitch across cube faces)
};

// ---- host state -------------------------------------------------------------------------------

struct Attachment {
    bt_attachment_config cfg;
    AttachmentMeta meta;
    uint64_t tile_bytes = 0;
    void* level0 = nullptr;           // atlas_size x T x T texels
    std::vector<void*> mips;          // level k (k>=1): atlas_size x (T>>k)^2 texels; lazily allocated
    // written[layer] == 0: the layer still holds bt_atlas_create's zeros (a wgpu texture starts zeroed) — nothing has run on it, been
    // uploaded or loaded into it, and its device pointer has not been handed out (bt_atlas_attachment_storage marks every layer).  A split of a
    // job whose finest tiles are all unwritten takes "the previous value" of a no-data pixel (split.wgsl:34-42) as 0 without fetching it.
    mutable std::vector<uint8_t> written;  // (mutable: handing out the storage pointer of a const atlas counts as a write, bt_atlas_attachment_storage)
    void mark_written(uint32_t first, uint32_t count) const {
        for (uint64_t i = first; i < uint64_t(first) + count && i < written.size(); i++) written[i] = 1;
    }
};

// TileState of the atlas (tile_atlas.rs:260-277): `loading` = 0 is LoadingState::Loaded, n > 0 is Loading(n).
struct TileState {
    uint32_t atlas_index;
    uint32_t requests;
    uint32_t loading;
};

// AtlasTileAttachment (tile_atlas.rs:62-67)
struct AtlasTileAttachment {
    bt_tile_coordinate coordinate;
    uint32_t atlas_index;
    uint32_t attachment_index;
};

enum TaskType : uint32_t { kSplit = 0, kStitch = 1, kDownsample = 2, kSave = 3, kBarrier = 4 };

struct Task {
    TaskType type;
    bt_tile_coordinate coord;
    uint32_t atlas_index;
    uint32_t attachment_index;
    bt_atlas_tile rel[8];
    float tl[2], br[2];
    int32_t raster;  // index into bt_preprocessor::rasters
    uint32_t job;    // which preprocess_tile / preprocess_spherical call queued it
};

struct Raster {
    RasterDev dev;
    uint32_t format;
    bool owned;
    // bt_raster.on_device == BT_RASTER_HOST_DEFERRED: the device buffer exists, the caller's rows are copied when the queue runs
    // (all at once by bt_preprocessor_run, band by band next to the kernels by bt_preprocessor_run_streamed)
    const void* host = nullptr;
    uint64_t host_bytes = 0;
    uint64_t host_pitch = 0;  // the caller's row pitch (the device copy's may be padded to 16 bytes)
    bool pending = false;
    const void* dev_src = nullptr;  // a borrowed device raster that is not 16-byte aligned: copied into the padded buffer `dev` by the first run
    uint64_t dev_src_pitch = 0;
    // the rectangles {x0, y0, x1, y1} of a deferred raster that have travelled (a sharded run: this rank's window): the device holds each of
    // them completely; a window counts as present when ONE of them contains it
    std::vector<std::array<uint32_t, 4>> windows;
    bool holds(const uint32_t w[4]) const {
        for (const std::array<uint32_t, 4>& h : windows)
            if (w[0] >= h[0] && w[1] >= h[1] && w[2] <= h[2] && w[3] <= h[3]) return true;
        return false;
    }
    void add_window(const uint32_t w[4]) {
        if (holds(w)) return;
        windows.erase(std::remove_if(windows.begin(), windows.end(), [&](const std::array<uint32_t, 4>& h) { return h[0] >= w[0] && h[1] >= w[1] && h[2] <= w[2] && h[3] <= w[3]; }),
                      windows.end());
        windows.push_back({w[0], w[1], w[2], w[3]});
    }
    uint64_t alloc_bytes = 0;  // size of the owned device allocation
};

}  // namespace bt

struct bt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    // pinned staging buffers of the tile save / load paths: allocated on first use (pinning 100 MB costs tens of
    // milliseconds), kept for the life of the context
    static constexpr uint32_t kStagingBuffers = 3;
    void* staging[kStagingBuffers] = {};
    size_t staging_bytes = 0;
    // streamed runs: uploads and downloads on their own queues beside the kernels' stream (created on first use)
    hipStream_t copy_stream = nullptr, save_stream = nullptr;
    // released raster allocations are kept for the next queue (hipMalloc + hipFree of a 512 MB source cost ~1.5 ms of the
    // end-to-end span; a host that preprocesses dataset after dataset pays them once; a cube job has six): at most kSpareRasters buffers
    // and kSpareRasterBytes in all, the smallest goes first (bt_ctx_trim gives them all back)
    static constexpr size_t kSpareRasters = 8;
    static constexpr uint64_t kSpareRasterBytes = 4ull << 30;
    std::vector<std::pair<void*, uint64_t>> spare_rasters;  // (device pointer, bytes)
    void* take_spare_raster(uint64_t need, uint64_t* bytes) {  // the smallest kept buffer that holds `need` bytes, or nullptr
        size_t best = spare_rasters.size();
        for (size_t i = 0; i < spare_rasters.size(); i++)
            if (spare_rasters[i].second >= need && (best == spare_rasters.size() || spare_rasters[i].second < spare_rasters[best].second)) best = i;
        if (best == spare_rasters.size()) return nullptr;
        void* ptr = spare_rasters[best].first;
        *bytes = spare_rasters[best].second;
        spare_rasters.erase(spare_rasters.begin() + long(best));
        return ptr;
    }
    void park_raster(void* ptr, uint64_t bytes) {  // keeps it, or frees what does not fit any more (the smallest first)
        spare_rasters.push_back({ptr, bytes});
        for (;;) {
            uint64_t total = 0;
            size_t smallest = 0;
            for (size_t i = 0; i < spare_rasters.size(); i++) {
                total += spare_rasters[i].second;
                if (spare_rasters[i].second < spare_rasters[smallest].second) smallest = i;
            }
            if (spare_rasters.size() <= kSpareRasters && total <= kSpareRasterBytes) break;
            hipFree(spare_rasters[smallest].first);
            spare_rasters.erase(spare_rasters.begin() + long(smallest));
        }
    }
    uint32_t io_threads = 0;  // writer / reader threads of the save and load paths; 0 = automatic (bt_ctx_set_io_threads)
};

namespace bt {
bt_status ctx_staging(bt_ctx* ctx, size_t bytes_per_buffer);  // ensures ctx->staging[*] hold at least that much
uint32_t usable_cpus();                  // CPUs this process may use: affinity mask capped by the cgroup CPU quota
uint32_t ctx_io_threads(const bt_ctx* ctx);  // the resolved thread count of the save / load paths
}

struct bt_atlas {
    bt_ctx* ctx = nullptr;
    bt_terrain_config config{};
    std::vector<bt::Attachment> attachments;
    // TileAtlasState (tile_atlas.rs:279-298)
    std::unordered_set<bt_tile_coordinate, bt::CoordHash, bt::CoordEq> existing_tiles;
    std::unordered_map<bt_tile_coordinate, bt::TileState, bt::CoordHash, bt::CoordEq> tile_states;
    std::deque<bt_atlas_tile> unused_tiles;        // LRU of free / released slots (:307-309, 459-476)
    std::deque<bt::AtlasTileAttachment> to_load;   // queued by request_tile (:445-451), drained by bt_atlas_update
    uint64_t state_version = 1;                    // bumped whenever tile_states changes (device copies are rebuilt lazily)
    // the Save tasks of the preprocessor runs since the last bt_preprocessor_save (preprocessor.rs:378-380)
    std::vector<bt::AtlasTileAttachment> to_save;
};

namespace bt {

// One launch of the compiled plan.
enum LaunchKind : uint32_t { kLaunchSplit, kLaunchDownsample, kLaunchStitch, kLaunchFusedMain, kLaunchFusedTail, kLaunchFusedDirect, kLaunchFusedTodo };
struct Launch {
    LaunchKind kind;
    uint32_t attachment;
    uint32_t first_task, task_count;  // into the device task array
    uint32_t aux0 = 0, aux1 = 0;
    uint64_t algorithmic_bytes = 0;  // inputs read once + outputs written once
    uint32_t phase = 0;              // sharded runs: 0 = BT_RUN_SHARD_LOCAL part, 2 = BT_RUN_SHARD_FINISH part
    uint32_t kernels = 1;            // kernels this plan entry launches (fused main without an LDS window: fused_corner + itself)
};

// host-side launchers implemented in bt_kernels.hip
bt_status launch_split(bt_ctx* ctx, const AttachmentMeta& m, void* atlas, const TaskDev* tasks, uint32_t n,
                       const RasterDev* rasters);
bt_status launch_downsample(bt_ctx* ctx, const AttachmentMeta& m, void* atlas, const TaskDev* tasks, uint32_t n);
// rows_only: just the top / bottom apron rows (full width, corners included) — the fused path writes the left / right
// apron columns of those tiles itself
bt_status launch_stitch(bt_ctx* ctx, const AttachmentMeta& m, void* atlas, const TaskDev* tasks, uint32_t n, bool rows_only = false, bool one_region = false);
bt_status launch_sample(bt_ctx* ctx, const AttachmentMeta& m, const void* atlas, const bt_tile_lookup* lookups, uint32_t count, float* out);
bt_status launch_mip_level(bt_ctx* ctx, uint32_t format, const void* parent, void* child, uint32_t parent_size,
                           uint32_t layers);
bt_status launch_gather_layers(hipStream_t stream, const void* atlas, const uint32_t* layers, uint32_t count, void* pinned_dst, uint64_t tile_bytes);
bt_status launch_synth_fbm(bt_ctx* ctx, void* dst, uint32_t w, uint32_t h, uint64_t pitch, uint32_t x0, uint32_t y0,
                           uint32_t base_cell, uint32_t octaves, uint32_t seed);

// fused-path state of a preprocessor (bt_fused.hip): launch descriptors + device buffers of its compiled queue
struct FusedState;
void fused_release(struct ::bt_preprocessor* p);

bt_status release_queue(struct ::bt_preprocessor* p);  // bt_run.cpp
bt_status ensure_compiled(struct ::bt_preprocessor* p, struct ::bt_atlas* a, uint32_t mode);  // bt_run.cpp: queue -> launch plan
bt_status run_plan_entry(struct ::bt_preprocessor* p, struct ::bt_atlas* a, const Launch& l);  // bt_run.cpp: one launch of the plan
uint32_t fused_begin_run(struct ::bt_preprocessor* p, struct ::bt_atlas* a);  // bt_fused.hip: per run, before its launches (FusedArgs::prev_zero, Attachment::written)
bt_status upload_pending_rasters(struct ::bt_preprocessor* p, const std::vector<uint8_t>* skip = nullptr);  // bt_host.cpp: deferred host rasters, all at once (skip[i]: not raster i)

// streamed run (bt_host.cpp drives it): a fused main / direct launch cut into bands of whole tile rows
struct StreamBand {
    uint32_t item_begin, item_count;  // into the job's item list ((side, tile row, x) order)
    uint32_t tile_y_begin, tile_y_end;
    uint32_t side, raster;            // the cube side of the band's tiles and the source raster they read
    uint32_t source_row_end;          // the band's kernels read rows of that raster below this one (exclusive)
};
// true when plan entry `l` is a fused main / direct launch that can run band by band (tile_rows_per_band 0: automatic)
bool fused_stream_bands(struct ::bt_preprocessor* p, const Launch& l, uint32_t tile_rows_per_band, std::vector<StreamBand>* bands);
bt_status fused_launch_range(struct ::bt_preprocessor* p, struct ::bt_atlas* a, const Launch& l, uint32_t item_begin, uint32_t item_count);
// the finest tiles [item_begin, item_begin + item_count) of a fused main / direct launch, in launch order
struct FusedTile {
    bt_tile_coordinate coordinate;
    uint32_t atlas_index;
};
void fused_launch_tiles(const struct ::bt_preprocessor* p, const Launch& l, uint32_t item_begin, uint32_t item_count, std::vector<FusedTile>* out);

// coordinate math (bt_host.cpp)
void tile_children(bt_tile_coordinate c, bt_tile_coordinate out[4]);
void tile_neighbours(bt_tile_coordinate c, bool spherical, bt_tile_coordinate out[8]);

}  // namespace bt

struct bt_preprocessor {
    bt_ctx* ctx = nullptr;
    std::vector<bt::Task> queue;
    std::vector<bt::Raster> rasters;
    uint32_t jobs = 0;
    uint32_t shard_rank = 0, shard_world = 1;
    bool shard_distributed = false;  // the last sharded run kept the finest LOD on its owners (BT_RUN_SHARD_DISTRIBUTED)
    std::vector<bt_shard_range> shard_ranges;
    std::vector<bt_shard_piece> shard_pieces;
    // BT_RUN_SHARD_OVERLAP: the job's collective runs on the communicator's stream between these two events
    hipEvent_t shard_local_done = nullptr, shard_exchange_done = nullptr;
    bool shard_exchange_pending = false;  // bt_preprocessor_finish_sharded has to follow
    uint64_t uploaded_source_bytes = 0;   // bytes of the last deferred host raster that actually travelled (a sharded run: its window)
    // compiled plan (rebuilt when the queue changes)
    bool compiled = false;
    bool saves_recorded = false;  // the kept queue's Save tasks are already in the atlas's to_save list
    uint32_t compiled_flags = 0;
    std::vector<bt::Launch> plan;
    bt::TaskDev* tasks_dev = nullptr;
    size_t tasks_dev_cap = 0;
    std::vector<bt::TaskDev> tasks_host;  // the same records on the host (which layers a batched launch writes: Attachment::written)
    bt::RasterDev* rasters_dev = nullptr;
    size_t rasters_dev_cap = 0;
    bt_run_stats stats{};
    bt::FusedState* fused = nullptr;  // owned; freed by fused_release
    // BT_RUN_PROFILE: events[run * (plan.size() + 1) + i]; event 0 of a run precedes its first launch
    std::vector<hipEvent_t> events;
    std::vector<hipEvent_t> event_pool;  // events read by bt_preprocessor_profile, kept for the next profiled runs (no hipEventCreate inside a timed step)
    uint32_t profiled_runs = 0;
    std::vector<uint32_t> profiled_phases;  // per profiled run: BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_FINISH bits of the plan halves it executed
};

namespace bt {
// the finest LOD among the sharded pieces of an attachment (what BT_RUN_SHARD_DISTRIBUTED leaves on its owners)
inline uint32_t shard_finest_lod(const ::bt_preprocessor* p, uint32_t attachment) {
    uint32_t lod = 0;
    for (const bt_shard_piece& piece : p->shard_pieces)
        if (piece.attachment_index == attachment && piece.lod > lod) lod = piece.lod;
    return lod;
}
// the rank that holds tile `atlas_index` of `attachment` after a distributed run: the owner of its finest-LOD piece, or,
// for every other tile (all ranks hold those), atlas_index % world — one writer per file
inline uint32_t shard_holder(const ::bt_preprocessor* p, uint32_t attachment, uint32_t lod, uint32_t atlas_index) {
    if (lod == shard_finest_lod(p, attachment))
        for (const bt_shard_piece& piece : p->shard_pieces)
            if (piece.attachment_index == attachment && piece.lod == lod && atlas_index >= piece.first_layer && atlas_index < piece.first_layer + piece.layers)
                return piece.owner_rank;
    return atlas_index % p->shard_world;
}
}  // namespace bt

namespace bt {
// the three forms of the tiling prepass with the view's approximate_height optionally taken from device memory (bt_frame_update);
// form: 0 = bt_tiling_prepass_run, 1 = _run_unordered, 2 = _run_plain
bt_status tiling_prepass_enqueue(bt_tiling_prepass* t, const bt_view_state* view, const float* device_height, uint32_t form);
bt_ctx* tiling_prepass_ctx(const bt_tiling_prepass* t);  // the context (stream) its kernels run on
}  // namespace bt
