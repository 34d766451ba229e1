// The collective step of a sharded preprocess job, issued by the library on the context's stream (new
// design, the reference is single-GPU).  RCCL is resolved at run time: the instance already loaded in the process
// (torch ships one; a Rust host links one) or librccl.so.1 — so the library has no link-time dependency on it and loads
// on machines without RCCL.
#include <dlfcn.h>

#include <cstring>
#include <vector>

#include "bt_internal.hpp"

using namespace bt;

namespace {

struct UniqueId {
    char internal[BT_COMM_UNIQUE_ID_BYTES];  // ncclUniqueId
};
typedef void* Comm;  // ncclComm_t
constexpr int kNcclUint8 = 1;  // ncclDataType_t::ncclUint8

struct Rccl {
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*AllGather)(const void*, void*, size_t, int, Comm, hipStream_t) = nullptr;
    int (*Broadcast)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

const Rccl& rccl() {
    static Rccl r = [] {
        Rccl t;
        // RTLD_DEFAULT is ((void*)0) on glibc: "found in the process" needs its own flag, not a non-null handle.  When RCCL
        // is already visible (a host that links it — the bt_comm_adopt case) its symbols MUST come from that instance,
        // which is what the default search order gives.
        void* handle = RTLD_DEFAULT;
        bool found = dlsym(RTLD_DEFAULT, "ncclAllGather") != nullptr;
        if (!found) {
            for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"})
                if ((handle = dlopen(name, RTLD_NOW | RTLD_GLOBAL))) {
                    found = true;
                    break;
                }
        }
        if (!found) return t;
        auto sym = [&](const char* n) { return dlsym(handle, n); };
        t.GetUniqueId = (decltype(t.GetUniqueId))sym("ncclGetUniqueId");
        t.CommInitRank = (decltype(t.CommInitRank))sym("ncclCommInitRank");
        t.CommDestroy = (decltype(t.CommDestroy))sym("ncclCommDestroy");
        t.GroupStart = (decltype(t.GroupStart))sym("ncclGroupStart");
        t.GroupEnd = (decltype(t.GroupEnd))sym("ncclGroupEnd");
        t.AllGather = (decltype(t.AllGather))sym("ncclAllGather");
        t.Broadcast = (decltype(t.Broadcast))sym("ncclBroadcast");
        t.GetErrorString = (decltype(t.GetErrorString))sym("ncclGetErrorString");
        t.ok = t.GetUniqueId && t.CommInitRank && t.CommDestroy && t.GroupStart && t.GroupEnd && t.AllGather && t.Broadcast;
        return t;
    }();
    return r;
}

bt_status need_rccl() {
    if (rccl().ok) return BT_OK;
    set_error("RCCL not available (no ncclAllGather in the process and librccl.so.1 not loadable)");
    return BT_ERR_UNSUPPORTED;
}

bt_status nccl_fail(int rc, const char* what) {
    set_error("RCCL error %d (%s) in %s", rc, rccl().GetErrorString ? rccl().GetErrorString(rc) : "?", what);
    return BT_ERR_DEVICE;
}

#define BT_NCCL(expr)                                \
    do {                                             \
        const int _rc = (expr);                      \
        if (_rc != 0) return nccl_fail(_rc, #expr);  \
    } while (0)

}  // namespace

struct bt_comm {
    bt_ctx* ctx = nullptr;
    Comm comm = nullptr;
    uint32_t world = 1, rank = 0;
    bool owned = false;
    hipStream_t stream = nullptr;  // BT_RUN_SHARD_OVERLAP: the collectives' own queue (created on first use)
};

extern "C" {

bt_status bt_comm_unique_id(uint8_t out[BT_COMM_UNIQUE_ID_BYTES]) {
    if (!out) return BT_ERR_INVALID_ARGUMENT;
    if (bt_status s = need_rccl()) return s;
    UniqueId id;
    BT_NCCL(rccl().GetUniqueId(&id));
    memcpy(out, id.internal, BT_COMM_UNIQUE_ID_BYTES);
    return BT_OK;
}

bt_status bt_comm_create(bt_ctx* ctx, uint32_t world, uint32_t rank, const uint8_t unique_id[BT_COMM_UNIQUE_ID_BYTES], bt_comm** out) {
    if (!ctx || !unique_id || !out || world == 0 || rank >= world) return BT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (bt_status s = need_rccl()) return s;
    BT_HIP(hipSetDevice(ctx->device));
    UniqueId id;
    memcpy(id.internal, unique_id, BT_COMM_UNIQUE_ID_BYTES);
    Comm c = nullptr;
    BT_NCCL(rccl().CommInitRank(&c, int(world), id, int(rank)));
    bt_comm* comm = new bt_comm();
    comm->ctx = ctx;
    comm->comm = c;
    comm->world = world;
    comm->rank = rank;
    comm->owned = true;
    *out = comm;
    return BT_OK;
}

bt_status bt_comm_adopt(bt_ctx* ctx, void* nccl_comm, uint32_t world, uint32_t rank, bt_comm** out) {
    if (!ctx || !nccl_comm || !out || world == 0 || rank >= world) return BT_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    if (bt_status s = need_rccl()) return s;
    bt_comm* comm = new bt_comm();
    comm->ctx = ctx;
    comm->comm = nccl_comm;
    comm->world = world;
    comm->rank = rank;
    *out = comm;
    return BT_OK;
}

void bt_comm_destroy(bt_comm* comm) {
    if (!comm) return;
    if (comm->stream) {
        hipStreamSynchronize(comm->stream);
        hipStreamDestroy(comm->stream);
    }
    if (comm->owned && comm->comm && rccl().ok) rccl().CommDestroy(comm->comm);
    delete comm;
}

// A health check of the communicator on the context's stream: every rank fills its slot of a device buffer with a rank pattern,
// ONE grouped collective (in-place all-gather + in-place broadcast from the last rank — the two shapes a sharded step issues) moves
// them, and every byte is verified on the host.  bt_comm_check: 4 KB slots; bt_comm_preflight: the caller's slot size (an atlas tile).
bt_status bt_comm_preflight(bt_comm* comm, uint64_t slot_bytes, float* elapsed_ms) {
    // (a health check, not a bandwidth test: a slot is a tile or a few — world + 1 slots are allocated on the device AND on the host)
    if (!comm || slot_bytes == 0 || slot_bytes > (64ull << 20)) {
        set_error("bt_comm_preflight: slot_bytes %llu (1 .. 64 MiB)", (unsigned long long)slot_bytes);
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (bt_status s = need_rccl()) return s;
    bt_ctx* ctx = comm->ctx;
    BT_HIP(hipSetDevice(ctx->device));
    const size_t slot = size_t(slot_bytes), total = slot * (comm->world + 1);
    uint8_t* dev = nullptr;
    BT_HIP(hipMalloc((void**)&dev, total));
    std::vector<uint8_t> host(total, 0);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bt_status rc = BT_OK;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipMemsetAsync(dev, 0, total, ctx->stream);
    if (e == hipSuccess) e = hipMemsetAsync(dev + slot * comm->rank, int(0x40 + comm->rank), slot, ctx->stream);
    if (e == hipSuccess && comm->rank == comm->world - 1) e = hipMemsetAsync(dev + slot * comm->world, 0x7E, slot, ctx->stream);
    if (e == hipSuccess) e = hipEventRecord(e0, ctx->stream);
    if (e == hipSuccess) {
        const Rccl& R = rccl();
        int r0 = R.GroupStart();
        int r1 = R.AllGather(dev + slot * comm->rank, dev, slot, kNcclUint8, comm->comm, ctx->stream);
        int r2 = R.Broadcast(dev + slot * comm->world, dev + slot * comm->world, slot, kNcclUint8, int(comm->world - 1), comm->comm, ctx->stream);
        int r3 = R.GroupEnd();
        if (r0 || r1 || r2 || r3) rc = nccl_fail(r0 ? r0 : r1 ? r1 : r2 ? r2 : r3, "bt_comm_preflight collective");
    }
    if (e == hipSuccess && rc == BT_OK) e = hipEventRecord(e1, ctx->stream);
    if (e == hipSuccess && rc == BT_OK) e = hipMemcpyAsync(host.data(), dev, total, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess && rc == BT_OK) e = hipStreamSynchronize(ctx->stream);
    float ms = 0.0f;
    if (e == hipSuccess && rc == BT_OK) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    hipFree(dev);
    if (e != hipSuccess) return hip_fail(e, "bt_comm_preflight");
    if (rc) return rc;
    for (uint32_t r = 0; r <= comm->world; r++) {
        const uint8_t want = r == comm->world ? 0x7E : uint8_t(0x40 + r);
        const uint8_t* got = host.data() + slot * r;
        if (got[0] == want && (slot == 1 || memcmp(got, got + 1, slot - 1) == 0)) continue;  // every byte equals the first, the first is right
        size_t i = 0;
        while (i < slot && got[i] == want) i++;
        set_error("bt_comm_preflight: rank %u of %u: slot %u byte %zu holds 0x%02x after the grouped all-gather + broadcast", comm->rank, comm->world, r, i, got[i]);
        return BT_ERR_DEVICE;
    }
    if (elapsed_ms) *elapsed_ms = ms;
    return BT_OK;
}

bt_status bt_comm_check(bt_comm* comm) { return bt_comm_preflight(comm, 4096, nullptr); }

bt_status bt_preprocessor_shard_pieces(const bt_preprocessor* p, bt_shard_piece* out, uint32_t cap, uint32_t* count) {
    if (!p || !count) return BT_ERR_INVALID_ARGUMENT;
    *count = uint32_t(p->shard_pieces.size());
    for (uint32_t i = 0; i < *count && i < cap && out; i++) out[i] = p->shard_pieces[i];
    return BT_OK;
}

}  // extern "C"

namespace {
// the exchange of one step as ONE grouped collective on `stream`: an in-place all-gather per LOD for the regular (planar)
// layout, an in-place broadcast per piece otherwise
bt_status grouped_exchange(bt_preprocessor* p, bt_atlas* a, bt_comm* comm, hipStream_t stream, bool distributed) {
    const Rccl& R = rccl();
    // BT_RUN_SHARD_DISTRIBUTED: the finest LOD of every attachment stays where it was computed
    auto stays = [&](uint32_t attachment, uint32_t lod) { return distributed && lod == shard_finest_lod(p, attachment); };
    BT_NCCL(R.GroupStart());
    int rc = 0;
    if (!p->shard_ranges.empty()) {
        for (const bt_shard_range& r : p->shard_ranges) {
            if (stays(r.attachment_index, r.lod)) continue;
            const Attachment& at = a->attachments[r.attachment_index];
            uint8_t* base = (uint8_t*)at.level0 + at.tile_bytes * r.first_layer;
            const size_t count = size_t(at.tile_bytes) * r.layers_per_rank;
            if (!rc) rc = R.AllGather(base + count * comm->rank, base, count, kNcclUint8, comm->comm, stream);
        }
    } else {
        for (const bt_shard_piece& piece : p->shard_pieces) {
            if (stays(piece.attachment_index, piece.lod)) continue;
            const Attachment& at = a->attachments[piece.attachment_index];
            uint8_t* buf = (uint8_t*)at.level0 + at.tile_bytes * piece.first_layer;
            if (!rc) rc = R.Broadcast(buf, buf, size_t(at.tile_bytes) * piece.layers, kNcclUint8, int(piece.owner_rank), comm->comm, stream);
        }
    }
    const int end = R.GroupEnd();
    if (rc) return nccl_fail(rc, "grouped collective");
    if (end) return nccl_fail(end, "ncclGroupEnd");
    return BT_OK;
}

bt_status check_comm(const bt_preprocessor* p, const bt_comm* comm) {
    if (comm->ctx != p->ctx || comm->world != p->shard_world || comm->rank != p->shard_rank) {
        set_error("communicator (rank %u of %u) does not match set_shard(%u, %u) / the preprocessor's context", comm->rank, comm->world, p->shard_rank,
                  p->shard_world);
        return BT_ERR_INVALID_ARGUMENT;
    }
    return BT_OK;
}
}  // namespace

namespace bt {
bt_status shard_exchange(bt_preprocessor* p, bt_atlas* a, bt_comm* comm, hipStream_t stream, bool distributed) {
    if (!comm) {
        set_error("the exchange of a sharded step needs a communicator");
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (bt_status s = need_rccl()) return s;
    if (comm->world == 1) return BT_OK;
    return grouped_exchange(p, a, comm, stream, distributed);
}
bt_status shard_check_comm(const bt_preprocessor* p, const bt_comm* comm) { return check_comm(p, comm); }
}  // namespace bt

extern "C" {

bt_status bt_preprocessor_run_sharded(bt_preprocessor* p, bt_atlas* a, bt_comm* comm, uint32_t flags) {
    if (!p || !a || !comm) return BT_ERR_INVALID_ARGUMENT;
    if (bt_status s = check_comm(p, comm)) return s;
    const uint32_t pass = flags & (BT_RUN_GENERIC | BT_RUN_PROFILE);
    const bool local_only = (flags & BT_RUN_SHARD_LOCAL) && !(flags & BT_RUN_SHARD_FINISH);
    const bool exchange_only = (flags & BT_RUN_SHARD_EXCHANGE) != 0;  // timing: the grouped collective of the compiled plan alone
    const bool overlap = (flags & BT_RUN_SHARD_OVERLAP) != 0;
    if (exchange_only && (local_only || comm->world == 1 || (p->shard_ranges.empty() && p->shard_pieces.empty()))) {
        set_error("BT_RUN_SHARD_EXCHANGE needs a sharded queue that has run once (and no BT_RUN_SHARD_LOCAL)");
        return BT_ERR_INVALID_ARGUMENT;
    }
    if (overlap) {
        // The local phase on the context's stream, the collective behind it on the communicator's own stream, nothing else:
        // the caller goes on with ANOTHER job's local phase (a second atlas) and comes back with bt_preprocessor_finish_sharded.
        if (exchange_only || local_only || !(flags & BT_RUN_KEEP_QUEUE)) {
            set_error("BT_RUN_SHARD_OVERLAP: a whole step of a kept queue (BT_RUN_KEEP_QUEUE, neither BT_RUN_SHARD_LOCAL nor BT_RUN_SHARD_EXCHANGE)");
            return BT_ERR_INVALID_ARGUMENT;
        }
        if (p->shard_exchange_pending) {
            set_error("BT_RUN_SHARD_OVERLAP: the previous step of this preprocessor has not been finished (bt_preprocessor_finish_sharded)");
            return BT_ERR_INVALID_ARGUMENT;
        }
        BT_HIP(hipSetDevice(p->ctx->device));
        if (!comm->stream) BT_HIP(hipStreamCreateWithFlags(&comm->stream, hipStreamNonBlocking));
        if (!p->shard_local_done) BT_HIP(hipEventCreateWithFlags(&p->shard_local_done, hipEventDisableTiming));
        if (!p->shard_exchange_done) BT_HIP(hipEventCreateWithFlags(&p->shard_exchange_done, hipEventDisableTiming));
        const uint32_t local_flags = comm->world == 1 ? 0u : BT_RUN_SHARD_LOCAL;  // (a world of one: the whole plan is "local")
        if (bt_status s = bt_preprocessor_run(p, a, pass | BT_RUN_KEEP_QUEUE | local_flags | (flags & BT_RUN_SHARD_DISTRIBUTED))) return s;
        BT_HIP(hipEventRecord(p->shard_local_done, p->ctx->stream));
        BT_HIP(hipStreamWaitEvent(comm->stream, p->shard_local_done, 0));
        if (comm->world > 1)
            if (bt_status s = grouped_exchange(p, a, comm, comm->stream, (flags & BT_RUN_SHARD_DISTRIBUTED) != 0)) return s;
        BT_HIP(hipEventRecord(p->shard_exchange_done, comm->stream));
        p->shard_exchange_pending = true;
        return BT_OK;
    }
    if (comm->world == 1) {
        // a world of one: nothing to exchange; the sharded entry point still runs both halves
        if (bt_status s = bt_preprocessor_run(p, a, pass | BT_RUN_KEEP_QUEUE)) return s;
    } else {
        if (!exchange_only)
            if (bt_status s = bt_preprocessor_run(p, a, pass | BT_RUN_KEEP_QUEUE | BT_RUN_SHARD_LOCAL | (flags & BT_RUN_SHARD_DISTRIBUTED))) return s;
        if (!local_only) {
            if (bt_status s = grouped_exchange(p, a, comm, p->ctx->stream, (flags & BT_RUN_SHARD_DISTRIBUTED) != 0)) return s;
            if (!exchange_only)
                if (bt_status s = bt_preprocessor_run(p, a, (flags & (BT_RUN_GENERIC | BT_RUN_SHARD_DISTRIBUTED)) | BT_RUN_KEEP_QUEUE | BT_RUN_SHARD_FINISH)) return s;
        }
    }
    if (!(flags & BT_RUN_KEEP_QUEUE)) return release_queue(p);
    return BT_OK;
}

// The second half of a BT_RUN_SHARD_OVERLAP step: the context's stream waits for the job's collective, then the finishing
// kernels (cross-strip aprons, the top LODs, cube seams) run.  Between the two calls the context's stream is free for the local
// phase of other jobs — that is the overlap: per step max(kernels, collective) instead of their sum.
bt_status bt_preprocessor_finish_sharded(bt_preprocessor* p, bt_atlas* a, bt_comm* comm, uint32_t flags) {
    if (!p || !a || !comm) return BT_ERR_INVALID_ARGUMENT;
    if (bt_status s = check_comm(p, comm)) return s;
    if (!p->shard_exchange_pending) {
        set_error("bt_preprocessor_finish_sharded: no BT_RUN_SHARD_OVERLAP step is pending");
        return BT_ERR_INVALID_ARGUMENT;
    }
    BT_HIP(hipSetDevice(p->ctx->device));
    BT_HIP(hipStreamWaitEvent(p->ctx->stream, p->shard_exchange_done, 0));
    p->shard_exchange_pending = false;
    if (comm->world > 1)
        if (bt_status s = bt_preprocessor_run(p, a, (flags & (BT_RUN_GENERIC | BT_RUN_SHARD_DISTRIBUTED | BT_RUN_PROFILE)) | BT_RUN_KEEP_QUEUE | BT_RUN_SHARD_FINISH)) return s;
    if (!(flags & BT_RUN_KEEP_QUEUE)) return release_queue(p);
    return BT_OK;
}

}  // extern "C"
