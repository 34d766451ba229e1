// bt_preprocessor_run: compiles the task queue into a launch plan and enqueues it on the stream.
//
// The reference drains its queue at most 32 tasks per attachment per frame, with a per-task uniform
// buffer + bind group, four tile copies per task and a one-frame readback (select_ready_tasks
// preprocessor.rs:346-399, GpuPreprocessor::prepare gpu_preprocessor.rs:120-223,
// TerrainPreprocessNode::run preprocess/mod.rs:143-218).  Here a phase (the tasks between two
// Barrier tasks) becomes ONE batched launch per task type, reading a device array of task records;
// where a job qualifies, split + the top of the LOD pyramid + aprons are fused (bt_fused.hip).
#include <algorithm>
#include <cstring>

#include "bt_internal.hpp"

using namespace bt;

namespace bt {
bool fused_plan(bt_preprocessor* p, bt_atlas* a, std::vector<TaskDev>& tasks, std::vector<Launch>& plan);
bt_status fused_launch(bt_preprocessor* p, bt_atlas* a, const Launch& l);
}  // namespace bt

namespace {

constexpr uint32_t kMaxProfiledRuns = 512;  // events kept until bt_preprocessor_profile reads them

TaskDev to_device_task(const Task& t) {
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
    if (t.type == kDownsample)
        for (int i = 4; i < 8; i++) d.rel_index[i] = BT_INVALID_ATLAS_INDEX;
    return d;
}

// reference-shaped plan: every maximal run of same-type, same-attachment tasks inside a phase is a launch
void generic_plan(const bt_preprocessor* p, const bt_atlas* a, std::vector<TaskDev>& tasks, std::vector<Launch>& plan) {
    size_t i = 0;
    const std::vector<Task>& q = p->queue;
    while (i < q.size()) {
        const Task& t = q[i];
        if (t.type == kBarrier || t.type == kSave) {
            i++;
            continue;
        }
        size_t j = i;
        const uint32_t first = uint32_t(tasks.size());
        while (j < q.size()) {
            // stitch phases of consecutive LODs are independent of each other (a stitch reads same-LOD centres and
            // writes its own apron): the barriers / saves between them do not end the launch
            if (t.type == kStitch && (q[j].type == kBarrier || q[j].type == kSave)) {
                j++;
                continue;
            }
            if (q[j].type != t.type || q[j].attachment_index != t.attachment_index) break;
            tasks.push_back(to_device_task(q[j]));
            j++;
        }
        Launch l{};
        l.kind = t.type == kSplit ? kLaunchSplit : t.type == kDownsample ? kLaunchDownsample : kLaunchStitch;
        l.attachment = t.attachment_index;
        l.first_task = first;
        l.task_count = uint32_t(tasks.size()) - first;
        {   // algorithmic bytes of this launch
            const AttachmentMeta& m = a->attachments[t.attachment_index].meta;
            const uint64_t bpp = m.pixel_size, T = m.texture_size, c = m.center_size, b = m.border_size;
            if (t.type == kSplit) {
                std::vector<bool> seen(p->rasters.size(), false);
                for (size_t k = i; k < j; k++)
                    if (q[k].raster >= 0 && !seen[q[k].raster]) {
                        seen[q[k].raster] = true;
                        l.algorithmic_bytes += uint64_t(p->rasters[q[k].raster].dev.width) * p->rasters[q[k].raster].dev.height * bpp;
                    }
                l.algorithmic_bytes += uint64_t(j - i) * T * T * bpp;
            } else if (t.type == kDownsample) {
                l.algorithmic_bytes = uint64_t(j - i) * (4 * c * c + T * T) * bpp;  // 4 child texels per centre pixel
            } else {
                l.algorithmic_bytes = uint64_t(l.task_count) * 2 * (2 * b * (T + c)) * bpp;  // apron read + written
            }
        }
        plan.push_back(l);
        i = j;
    }
}

bt_status ensure_device_array(void** ptr, size_t* cap, size_t bytes) {
    if (*cap >= bytes && *ptr) return BT_OK;
    if (*ptr) hipFree(*ptr);
    *ptr = nullptr;
    *cap = 0;
    BT_HIP(hipMalloc(ptr, bytes ? bytes : 1));
    *cap = bytes;
    return BT_OK;
}

}  // namespace

namespace bt {
// queue -> launch plan (once per queue and mode)
bt_status ensure_compiled(bt_preprocessor* p, bt_atlas* a, uint32_t mode) {
    if (mode & BT_RUN_REFERENCE_DISPATCH) {  // only attachments with T % 8 != 0 behave differently: without one in the queue the flag is dropped
        bool odd = false;
        for (const Task& t : p->queue) odd = odd || a->attachments[t.attachment_index].meta.texture_size % 8u != 0;
        if (!odd) mode &= ~uint32_t(BT_RUN_REFERENCE_DISPATCH);
    }
    if (!p->compiled || p->compiled_flags != mode) {
        std::vector<TaskDev> tasks;
        p->plan.clear();
        p->shard_ranges.clear();
        p->shard_pieces.clear();
        bool fused = false;
        if (!mode) fused = fused_plan(p, a, tasks, p->plan);
        if (!fused) {
            fused_release(p);  // (a plan that was refused half way, or the fused plan of an earlier mode: bt_preprocessor_source_window must not see its jobs)
            tasks.clear();
            p->plan.clear();
            generic_plan(p, a, tasks, p->plan);
        }
        std::vector<RasterDev> rasters;
        for (const Raster& r : p->rasters) rasters.push_back(r.dev);
        if (bt_status s = ensure_device_array((void**)&p->tasks_dev, &p->tasks_dev_cap, tasks.size() * sizeof(TaskDev))) return s;
        if (bt_status s = ensure_device_array((void**)&p->rasters_dev, &p->rasters_dev_cap, rasters.size() * sizeof(RasterDev))) return s;
        // synchronous staging of two small arrays; done once per queue, not per run
        if (!tasks.empty()) BT_HIP(hipMemcpy(p->tasks_dev, tasks.data(), tasks.size() * sizeof(TaskDev), hipMemcpyHostToDevice));
        if (!rasters.empty()) BT_HIP(hipMemcpy(p->rasters_dev, rasters.data(), rasters.size() * sizeof(RasterDev), hipMemcpyHostToDevice));
        p->tasks_host = tasks;

        // statistics: algorithmic bytes = every source texel once + every produced tile texel once
        bt_run_stats st{};
        std::vector<bool> raster_used(p->rasters.size(), false);
        for (const Task& t : p->queue) {
            if (t.type == kSplit && t.raster >= 0 && !raster_used[t.raster]) {
                raster_used[t.raster] = true;
                const Raster& r = p->rasters[t.raster];
                st.algorithmic_bytes += uint64_t(r.dev.width) * r.dev.height * (r.format == BT_FORMAT_R16 ? 2 : 4);
            }
            if (t.type == kStitch) {  // every tile of a job is stitched exactly once
                st.tiles++;
                st.algorithmic_bytes += a->attachments[t.attachment_index].tile_bytes;
            }
        }
        st.kernel_launches = 0;
        for (const Launch& l : p->plan) st.kernel_launches += l.kernels;
        st.fused_jobs = fused ? p->jobs : 0;
        st.generic_jobs = fused ? 0 : p->jobs;
        p->stats = st;
        p->compiled = true;
        p->compiled_flags = mode;
        p->event_pool.insert(p->event_pool.end(), p->events.begin(), p->events.end());
        p->events.clear();
        p->profiled_runs = 0;
        p->profiled_phases.clear();
    }
    return BT_OK;
}

bt_status run_plan_entry(bt_preprocessor* p, bt_atlas* a, const Launch& l) {
    const Attachment& at = a->attachments[l.attachment];
    AttachmentMeta meta = at.meta;
    if (p->compiled_flags & BT_RUN_REFERENCE_DISPATCH) meta.row_limit = meta.texture_size / 8u * 8u;  // (gpu_tile_atlas.rs:105)
    switch (l.kind) {
        case kLaunchSplit:
            return launch_split(p->ctx, meta, at.level0, p->tasks_dev + l.first_task, l.task_count, p->rasters_dev);
        case kLaunchDownsample:
            return launch_downsample(p->ctx, meta, at.level0, p->tasks_dev + l.first_task, l.task_count);
        case kLaunchStitch:
            return launch_stitch(p->ctx, meta, at.level0, p->tasks_dev + l.first_task, l.task_count, l.aux0 == 1u, l.aux0 == 2u);
        default:
            return fused_launch(p, a, l);
    }
}
}  // namespace bt

extern "C" bt_status bt_preprocessor_run(bt_preprocessor* p, bt_atlas* a, uint32_t flags) {
    if (!p || !a) return BT_ERR_INVALID_ARGUMENT;
    if (p->ctx != a->ctx) {
        set_error("preprocessor and atlas belong to different contexts");
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(p->ctx->device));
    const uint32_t mode = flags & (BT_RUN_GENERIC | BT_RUN_REFERENCE_DISPATCH);
    if (bt_status s = ensure_compiled(p, a, mode)) return s;
    if ((flags & BT_RUN_PROFILE) != 0 && p->profiled_runs >= kMaxProfiledRuns) {
        set_error("BT_RUN_PROFILE: %u profiled runs are pending; read them with bt_preprocessor_profile() first", kMaxProfiledRuns);
        return BT_ERR_INVALID_ARGUMENT;
    }
    // every argument check comes before the first profile event and before the atlas bookkeeping: an early error return leaves neither an
    // orphan event (the rows of p->events would no longer line up with profiled_phases) nor layers marked written by launches that never ran
    const bool sharded = p->shard_world > 1;
    if (sharded && (flags & BT_RUN_SHARD_DISTRIBUTED)) {
        // the finest LOD stays on its owners: possible when nothing after the exchange reads finest tiles of other
        // ranks, i.e. not for cube jobs (their face seams are stitched from the neighbour face's finest tiles)
        for (const bt_shard_piece& piece : p->shard_pieces)
            if (piece.side != p->shard_pieces[0].side) {
                set_error("BT_RUN_SHARD_DISTRIBUTED needs a one-sided (planar) job: cube seams read finest tiles of other ranks");
                return BT_ERR_UNSUPPORTED;
            }
    }
    if (sharded && !(flags & (BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_FINISH))) {
        set_error("a sharded preprocessor runs with BT_RUN_SHARD_LOCAL and / or BT_RUN_SHARD_FINISH");
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (sharded) p->shard_distributed = (flags & BT_RUN_SHARD_DISTRIBUTED) != 0 && !p->shard_pieces.empty();
    if (bt_status s = upload_pending_rasters(p)) return s;  // rasters handed over with BT_RASTER_HOST_DEFERRED (sharded: this rank's window of them)

    const bool profile = (flags & BT_RUN_PROFILE) != 0;
    auto record = [&](void) -> bt_status {
        hipEvent_t e;
        if (!p->event_pool.empty()) {  // (profiled steps sit inside the bench's timed region: the events are created once and recycled)
            e = p->event_pool.back();
            p->event_pool.pop_back();
        } else {
            BT_HIP(hipEventCreate(&e));
        }
        p->events.push_back(e);
        BT_HIP(hipEventRecord(e, p->ctx->stream));
        return BT_OK;
    };
    if (profile)
        if (bt_status s = record()) return s;
    // (a sharded step's finishing half alone launches no split: the flags of its local half stand)
    if (!sharded || (flags & BT_RUN_SHARD_LOCAL)) p->stats.prev_zero_launches = fused_begin_run(p, a);
    for (const Launch& l : p->plan) {
        if (sharded && !(flags & (l.phase == 2 ? BT_RUN_SHARD_FINISH : BT_RUN_SHARD_LOCAL))) {
            if (profile)
                if (bt_status s2 = record()) return s2;
            continue;
        }
        if (bt_status s = run_plan_entry(p, a, l)) return s;
        if (profile)
            if (bt_status s2 = record()) return s2;
    }
    if (profile) {
        // which halves of the plan this run executed: a sharded step is two profiled runs (local, finish) that each leave a whole event row;
        // bt_preprocessor_profile averages a launch over the rows that ran it
        p->profiled_phases.push_back(sharded ? (flags & (BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_FINISH)) : uint32_t(BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_FINISH));
        p->profiled_runs++;
    }

    // Save tasks: remembered until bt_preprocessor_save (the reference starts them as tasks drain)
    if (!p->saves_recorded) {  // (re-runs of a kept queue produce the same tiles: recorded once per queue and save)
        for (const Task& t : p->queue)
            if (t.type == kSave) a->to_save.push_back({t.coord, t.atlas_index, t.attachment_index});
        p->saves_recorded = true;
    }
    if (!(flags & BT_RUN_KEEP_QUEUE)) return release_queue(p);
    return BT_OK;
}

namespace bt {
// the queue has run: drop it (and the rasters it uploaded) so that the next preprocess_* call starts afresh
bt_status release_queue(bt_preprocessor* p) {
    BT_HIP(hipStreamSynchronize(p->ctx->stream));  // borrowed rasters may be released by the caller afterwards
    p->queue.clear();
    for (Raster& r : p->rasters)
        if (r.owned && r.dev.data) p->ctx->park_raster((void*)r.dev.data, r.alloc_bytes);  // kept for the next queue's rasters (bt_ctx::spare_rasters)
    p->rasters.clear();
    p->jobs = 0;
    p->compiled = false;
    p->saves_recorded = false;
    return BT_OK;
}
}  // namespace bt

extern "C" bt_status bt_preprocessor_profile(bt_preprocessor* p, bt_launch_profile* out, uint32_t cap, uint32_t* count) {
    if (!p || !count || (!out && cap)) return BT_ERR_INVALID_ARGUMENT;
    const uint32_t n = uint32_t(p->plan.size());
    *count = n;
    BT_HIP(hipStreamSynchronize(p->ctx->stream));
    for (uint32_t i = 0; i < n && i < cap; i++) {
        const Launch& l = p->plan[i];
        double total = 0.0;
        uint32_t samples = 0;
        for (uint32_t r = 0; r < p->profiled_runs; r++) {
            // a run that skipped this launch (the other half of a sharded step) is not a sample of it
            const uint32_t ran = r < p->profiled_phases.size() ? p->profiled_phases[r] : uint32_t(BT_RUN_SHARD_LOCAL | BT_RUN_SHARD_FINISH);
            if (!(ran & (l.phase == 2 ? BT_RUN_SHARD_FINISH : BT_RUN_SHARD_LOCAL))) continue;
            float ms = 0.0f;
            BT_HIP(hipEventElapsedTime(&ms, p->events[size_t(r) * (n + 1) + i], p->events[size_t(r) * (n + 1) + i + 1]));
            total += ms;
            samples++;
        }
        out[i].kind = uint32_t(l.kind);
        out[i].tasks = l.task_count;
        out[i].algorithmic_bytes = l.algorithmic_bytes;
        out[i].samples = samples;
        out[i].avg_ms = samples ? float(total / samples) : 0.0f;
    }
    p->event_pool.insert(p->event_pool.end(), p->events.begin(), p->events.end());
    p->events.clear();
    p->profiled_runs = 0;
    p->profiled_phases.clear();
    return BT_OK;
}

namespace bt {
bool fused_source_window(const bt_preprocessor* p, uint32_t raster, uint32_t out[4]);
}

// Which texels of source raster `raster_index` (the order of the preprocess_* calls; a cube job adds six) does this preprocessor
// read?  Compiles the plan if necessary.  A sharded fused plan: this rank's column strips + halo; anything else: the whole raster.
extern "C" bt_status bt_preprocessor_source_window(bt_preprocessor* p, bt_atlas* a, uint32_t raster_index, uint32_t flags, uint32_t window[4], uint64_t* uploaded_bytes) {
    if (!p || !a || !window) return BT_ERR_INVALID_ARGUMENT;
    if (raster_index >= p->rasters.size()) {
        set_error("raster %u of %zu", raster_index, p->rasters.size());
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (bt_status s = ensure_compiled(p, a, flags & (BT_RUN_GENERIC | BT_RUN_REFERENCE_DISPATCH))) return s;
    if (!fused_source_window(p, raster_index, window)) {
        window[0] = window[1] = 0;
        window[2] = p->rasters[raster_index].dev.width;
        window[3] = p->rasters[raster_index].dev.height;
    }
    if (uploaded_bytes) *uploaded_bytes = p->uploaded_source_bytes;
    return BT_OK;
}

extern "C" bt_status bt_preprocessor_set_shard(bt_preprocessor* p, uint32_t rank, uint32_t world) {
    if (!p || world == 0 || rank >= world) return BT_ERR_INVALID_ARGUMENT;
    if (p->shard_rank != rank || p->shard_world != world) p->compiled = false;
    p->shard_rank = rank;
    p->shard_world = world;
    return BT_OK;
}

extern "C" bt_status bt_preprocessor_shard_ranges(const bt_preprocessor* p, bt_shard_range* out, uint32_t cap, uint32_t* count) {
    if (!p || !count) return BT_ERR_INVALID_ARGUMENT;
    *count = uint32_t(p->shard_ranges.size());
    for (uint32_t i = 0; i < *count && i < cap && out; i++) out[i] = p->shard_ranges[i];
    return BT_OK;
}
