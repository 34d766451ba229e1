/*
 * bevy_terrain_amd.h — C ABI of the MI355X-native terrain preprocessing + tiling-prepass backend.
 *
 * This is the drop-in boundary.  The reference (kurtkuehnert/bevy_terrain @ 2025-03-14) has no
 * FFI: its preprocessing and tile refinement run as wgpu compute passes inside a Bevy render
 * graph.  A Rust host keeps the public API (TerrainPlugin / TerrainPreprocessPlugin /
 * Preprocessor / TileAtlas / TileTree) and replaces the internal seam
 *     GpuPreprocessor::prepare + TerrainPreprocessNode::run + GpuAtlasAttachment::{copy_*,
 *     download_tiles, start_downloading_tiles}           (src/preprocess/mod.rs:143-218,
 *                                                         src/preprocess/gpu_preprocessor.rs:120-223,
 *                                                         src/terrain_data/gpu_tile_atlas.rs:276-412)
 *     TilingPrepassNode::run + TerrainViewData buffers   (src/render/tiling_prepass.rs:204-272,
 *                                                         src/render/terrain_view_bind_group.rs:118-247)
 * with calls into this library (see INTEGRATION.md for the `extern "C"` block).
 *
 * Conventions: opaque handles; POD structs with explicit layout; every function returns a
 * bt_status (0 = ok, < 0 = error) and never throws or aborts; bt_last_error() returns the text of
 * the last error of the calling thread; no callbacks; no global state besides that error string
 * (per-queue device buffers live in the bt_preprocessor that built them);
 * one bt_ctx per GPU and per host thread that drives it.  All file:line citations are relative to the reference checkout.
 */
#ifndef BEVY_TERRAIN_AMD_H
#define BEVY_TERRAIN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 6 (round 6): bt_preprocessor_run_streamed pipelines every fused job of a queue (several attachments, the six rasters of a cube job) and
 * bt_preprocessor_run_streamed_sharded does the same for a BT_RUN_SHARD_DISTRIBUTED rank; bt_stream_stats and bt_run_stats GREW (new
 * trailing fields: callers must be rebuilt against this header); a split onto layers nothing has written since bt_atlas_create takes
 * the "previous value" of a no-data pixel as 0 without fetching it (bt_run_stats.prev_zero_launches) — same bytes;
 * bt_atlas_attachment_storage now counts as a write to every layer of the attachment.  Two BEHAVIOUR CHANGES (not additions): a borrowed
 * unaligned device raster (on_device = 1) is copied into the library's padded buffer by EVERY run of a kept queue, not only the first
 * (round 5 froze it at the first run); and the rows of a BT_RASTER_HOST_DEFERRED raster must stay alive until the queue is RELEASED,
 * not merely until its first run (a kept queue that is re-planned for another rank / mode reads what the device does not hold yet).
 * 5 (round 5): + bt_ctx_set_io_threads / bt_ctx_io_threads (writer / reader threads of the save and load paths follow the CPUs the
 * process may use), bt_comm_preflight (tile-sized health check of the communicator before the first sharded step) — additions only.
 * 4 (round 4): + bt_frame_update / bt_frame_info, BT_RUN_SHARD_OVERLAP + bt_preprocessor_finish_sharded, bt_preprocessor_source_window, BT_RUN_REFERENCE_DISPATCH, bt_ctx_trim; the fused plan no longer has a fused_todo launch (kind 6 of bt_launch_profile
 * does not occur any more) — additions only.
 * 3 (round 3): + bt_preprocessor_run_streamed, BT_RASTER_HOST_DEFERRED, BT_RUN_SHARD_EXCHANGE, bt_tiling_prepass_run_plain /
 * _run_unordered / _set_window, launch kind 6 (fused todo) in bt_launch_profile — additions only, every version-2 call keeps its
 * meaning. */
#define BT_ABI_VERSION 6u

typedef int32_t bt_status;
enum {
    BT_OK = 0,
    BT_ERR_INVALID_ARGUMENT = -1,
    BT_ERR_ATLAS_OUT_OF_INDICES = -2, /* the reference panics "Atlas out of indices", tile_atlas.rs:384 */
    BT_ERR_DEVICE = -3,               /* a HIP call failed; text in bt_last_error() */
    BT_ERR_IO = -4,
    BT_ERR_UNSUPPORTED = -5, /* e.g. AttachmentFormat::Rgb8 / Rg16: silently unprocessed upstream */
    BT_ERR_OUT_OF_MEMORY = -6,
    BT_ERR_OVERFLOW = -7, /* tile buffer of the tiling prepass too small */
};

#define BT_INVALID_ATLAS_INDEX 0xFFFFFFFFu /* terrain_data/mod.rs:34 */
#define BT_INVALID_LOD 0xFFFFFFFFu         /* terrain_data/mod.rs:35 */
#define BT_MAX_ATTACHMENTS 8u              /* shaders/bindings.wgsl:16-31 */

/* AttachmentFormat (terrain_data/mod.rs:37-57); values are AttachmentFormat::id(). */
enum {
    BT_FORMAT_RGBA8 = 0,
    BT_FORMAT_R16 = 1,
    BT_FORMAT_RG16 = 3, /* accepted in configs, not processed (like the reference) -> BT_ERR_UNSUPPORTED */
    BT_FORMAT_RGB8 = 5, /* dito */
};

/* TileCoordinate (math/coordinate.rs:155-167) — 16 bytes, also the GPU layout (types.wgsl:25-29). */
typedef struct bt_tile_coordinate {
    uint32_t side, lod, x, y;
} bt_tile_coordinate;

/* AtlasTile (terrain_data/tile_atlas.rs:30-35) — 32 bytes, GPU layout preprocessing.wgsl:16-22. */
typedef struct bt_atlas_tile {
    bt_tile_coordinate coordinate;
    uint32_t atlas_index;
    uint32_t _padding[3];
} bt_atlas_tile;

/* AttachmentConfig (terrain_data/mod.rs:87-109). */
typedef struct bt_attachment_config {
    char name[64];
    uint32_t texture_size;    /* default 512 */
    uint32_t border_size;     /* default 1 */
    uint32_t mip_level_count; /* default 1 */
    uint32_t format;          /* BT_FORMAT_* ; default R16 */
} bt_attachment_config;

/* TerrainConfig (terrain.rs:26-49); only the fields the path reads. */
typedef struct bt_terrain_config {
    uint32_t lod_count;    /* default 1 */
    uint32_t atlas_size;   /* default 1024 */
    uint32_t spherical;    /* TerrainModel::is_spherical(), math/terrain_model.rs:54-60 */
    uint32_t attachment_count;
    bt_attachment_config attachments[BT_MAX_ATTACHMENTS];
    char path[256];        /* terrain folder below the assets root */
} bt_terrain_config;

/* A source raster handed to Preprocessor::preprocess_tile in place of asset_server.load(path)
 * (preprocessor.rs:240).  Texels: u16 (R16) or RGBA8, row-major, `row_pitch` bytes per row. */
typedef struct bt_raster {
    const void* data;
    uint32_t width, height;
    uint64_t row_pitch;  /* bytes; 0 = tightly packed */
    uint32_t format;     /* BT_FORMAT_R16 or BT_FORMAT_RGBA8; must equal the attachment's */
    uint32_t on_device;  /* 0: host memory (copied to the GPU by the call), 1: device pointer (borrowed
                            until the preprocessor has run and read at run time, on EVERY run of a kept queue; an R16
                            raster whose base or pitch is not a multiple of 16 bytes is copied device-to-device into a
                            padded buffer of the library's at the start of each run — up to width x height x 2 bytes of
                            device memory, 0.15 ms for 0.5 GB), BT_RASTER_HOST_DEFERRED: host memory that stays the
                            caller's until the queue is RELEASED — copied by bt_preprocessor_run (all at once) or by
                            bt_preprocessor_run_streamed (band by band, beside the kernels and the downloads).  A queue kept
                            with BT_RUN_KEEP_QUEUE may read the rows again (only the window a sharded rank needs travels;
                            a later bt_preprocessor_set_shard / other run flags fetch what the device does not hold yet) */
} bt_raster;
#define BT_RASTER_HOST_DEFERRED 2u

/* A decoded source image in host memory, ready to be handed over as a bt_raster (on_device = 0).  Replaces
 * `asset_server.load(path)` + preprocessor_load_tile (preprocessor.rs:240, 401-422; formats/tiff.rs:14-62): a 16-bit
 * grayscale PNG / TIFF decodes to R16 texels (host byte order), an 8-bit gray / gray + alpha / RGB / RGBA PNG or TIFF to Rgba8 (gray
 * replicated, alpha 255 where the file has none, like Bevy's Image::from_dynamic).  PNG: non-interlaced, 8 / 16 bit.  TIFF: classic
 * (II / MM), strips or tiles, uncompressed / LZW / deflate / PackBits, horizontal predictor.  Anything else:
 * BT_ERR_UNSUPPORTED.  `data` is owned by the library until bt_image_free. */
typedef struct bt_image {
    void* data;
    uint32_t width, height;
    uint32_t format;    /* BT_FORMAT_R16 or BT_FORMAT_RGBA8 */
    uint64_t row_pitch; /* bytes, tightly packed */
} bt_image;
bt_status bt_image_load(const char* path, uint32_t format, bt_image* out);
bt_status bt_image_decode(const void* bytes, size_t size, uint32_t format, bt_image* out);
void bt_image_free(bt_image* image);

/* PreprocessDataset (preprocessor.rs:35-55). */
typedef struct bt_preprocess_dataset {
    uint32_t attachment_index;
    uint32_t side;
    float top_left[2];     /* default (0,0) */
    float bottom_right[2]; /* default (1,1) */
    uint32_t lod_begin, lod_end; /* lod_range = lod_begin..lod_end, default 0..1 */
} bt_preprocess_dataset;

/* SphericalDataset (preprocessor.rs:29-33); rasters are passed next to it, one per cube side. */
typedef struct bt_spherical_dataset {
    uint32_t attachment_index;
    uint32_t lod_begin, lod_end;
} bt_spherical_dataset;

typedef struct bt_ctx bt_ctx;                   /* one GPU + one stream */
typedef struct bt_atlas bt_atlas;               /* TileAtlas + GpuTileAtlas */
typedef struct bt_preprocessor bt_preprocessor; /* Preprocessor + GpuPreprocessor */
typedef struct bt_tiling_prepass bt_tiling_prepass; /* TerrainViewData buffers + TilingPrepassNode */

/* ------------------------------------------------------------------ context */
uint32_t bt_abi_version(void);
const char* bt_last_error(void);
/* `stream` is a hipStream_t owned by the caller (NULL = the library creates its own). */
bt_status bt_ctx_create(int32_t device, void* stream, bt_ctx** out);
void bt_ctx_destroy(bt_ctx* ctx);
bt_status bt_ctx_set_stream(bt_ctx* ctx, void* stream);
void* bt_ctx_stream(const bt_ctx* ctx);
bt_status bt_ctx_synchronize(bt_ctx* ctx);
/* Gives back what the context keeps between queues: the device rasters finished queues released (kept so that the next queue's
 * sources need not be allocated again: 0.5 GB for a 16k R16 raster, six of 128 MB for a cube job; at most 8 buffers and 4 GiB) and
 * the pinned staging buffers of the save / load paths.
 * Synchronises the context's stream first.  `freed_bytes` (may be NULL): device + pinned bytes released. */
bt_status bt_ctx_trim(bt_ctx* ctx, uint64_t* freed_bytes);
/* Host threads that write (bt_preprocessor_save / _run_streamed) and read (bt_atlas_load_tiles) tile files for this context.
 * 0 = automatic: min(16, CPUs this process may use) — the affinity mask capped by the cgroup's CPU quota, not the machine's
 * hardware threads (the reference spawns one AsyncComputeTaskPool task per tile, tile_atlas.rs:77-116).  bt_ctx_io_threads returns
 * the number the next save / load will use. */
bt_status bt_ctx_set_io_threads(bt_ctx* ctx, uint32_t threads);
uint32_t bt_ctx_io_threads(const bt_ctx* ctx);
/* hipEvent pair on the context's stream: begin .. end -> elapsed milliseconds (end synchronises) */
bt_status bt_ctx_timer_begin(bt_ctx* ctx);
bt_status bt_ctx_timer_end(bt_ctx* ctx, float* elapsed_ms);
bt_status bt_device_malloc(bt_ctx* ctx, size_t bytes, void** out);
bt_status bt_device_free(bt_ctx* ctx, void* ptr);
bt_status bt_memcpy_h2d(bt_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
bt_status bt_memcpy_d2h(bt_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);

/* --------------------------------------------- TileCoordinate (coordinate.rs) */
void bt_tile_children(bt_tile_coordinate c, bt_tile_coordinate out[4]);                     /* :196-206 */
void bt_tile_neighbours(bt_tile_coordinate c, uint32_t spherical, bt_tile_coordinate out[8]); /* :208-279 */
bt_tile_coordinate bt_tile_parent(bt_tile_coordinate c);                                    /* :187-194 */
/* "{side}_{lod}_{x}_{y}" (:282-286); returns the length written (without NUL) */
int32_t bt_tile_name(bt_tile_coordinate c, char* buf, size_t cap);

/* ------------------------------------- TileAtlas (terrain_data/tile_atlas.rs) */
/* TileAtlas::new (:531-551): allocates atlas_size x T x T texels per attachment in HBM (zeroed,
 * like a fresh wgpu texture), the index allocator and the existing-tile set. */
bt_status bt_atlas_create(bt_ctx* ctx, const bt_terrain_config* config, bt_atlas** out);
void bt_atlas_destroy(bt_atlas* atlas);
/* TileAtlas::get_tile / get_or_allocate_tile (:553-559, 369-416). */
bt_status bt_atlas_get_tile(bt_atlas* atlas, bt_tile_coordinate c, bt_atlas_tile* out);
bt_status bt_atlas_get_or_allocate_tile(bt_atlas* atlas, bt_tile_coordinate c, bt_atlas_tile* out);
/* TileTreeEntry (terrain_data/tile_tree.rs:49-66) — 8 bytes, also the GPU layout (types.wgsl: TileTreeEntry). */
typedef struct bt_tile_tree_entry {
    uint32_t atlas_index; /* BT_INVALID_ATLAS_INDEX: nothing loaded */
    uint32_t atlas_lod;   /* BT_INVALID_LOD */
} bt_tile_tree_entry;
/* The streaming side of TileAtlasState (tile_atlas.rs:418-503): request_tile (a tile not present gets the oldest
 * unused slot and is queued for loading, one entry per attachment), release_tile (the last release puts the slot at
 * the back of the LRU; its data stays cached until the slot is reused), get_best_tile (the tile itself or its closest
 * loaded ancestor).  Releasing a tile that is not present is BT_ERR_INVALID_ARGUMENT (the reference panics). */
bt_status bt_atlas_request_tile(bt_atlas* atlas, bt_tile_coordinate c);
bt_status bt_atlas_release_tile(bt_atlas* atlas, bt_tile_coordinate c);
bt_status bt_atlas_get_best_tile(const bt_atlas* atlas, bt_tile_coordinate c, bt_tile_tree_entry* out);
/* TileAtlasState::update + AtlasAttachment::update (:327-345, 195-224), synchronously: starts and finishes up to
 * `max_loads` queued tile loads (0 = all): "{assets_root}/{config.path}/data/{name}/{coord}.bin" -> the tile's atlas
 * layer (+ its mip levels); a tile is Loaded once all its attachments are.  A missing / short file leaves the tile
 * Loading forever, like the reference (:202-204); *loaded / *failed (optional) count this call's outcomes. */
bt_status bt_atlas_update(bt_atlas* atlas, const char* assets_root, uint32_t max_loads, uint32_t* loaded, uint32_t* failed);
/* number of queued loads (to_load.len()) */
uint32_t bt_atlas_pending_loads(const bt_atlas* atlas);

/* existing_tiles in atlas-index (= allocation) order. Returns the tile count; fills up to `cap`. */
uint32_t bt_atlas_tiles(const bt_atlas* atlas, bt_tile_coordinate* coords, uint32_t* atlas_indices, uint32_t cap);
/* Device storage of one attachment: layer `i` starts at ptr + i*tile_bytes; rows are T*pixel_size
 * bytes, tightly packed, texel layout = the `.bin` tile file layout.  The caller may write through the pointer (a host-side
 * collective does): asking for it (device_ptr != NULL) counts as a write to every layer (bt_run_stats.prev_zero_launches). */
bt_status bt_atlas_attachment_storage(const bt_atlas* atlas, uint32_t attachment_index, void** device_ptr,
                                      uint64_t* tile_bytes, uint32_t* layers);
/* download + de-pad (gpu_tile_atlas.rs:338-412): `count` consecutive layers into host memory. */
bt_status bt_atlas_download_tiles(bt_atlas* atlas, uint32_t attachment_index, uint32_t first_layer,
                                  uint32_t count, void* dst_host, uint64_t dst_bytes);
/* upload_tiles level 0 (gpu_tile_atlas.rs:309-336). */
bt_status bt_atlas_upload_tile(bt_atlas* atlas, uint32_t attachment_index, uint32_t layer, const void* src_host,
                               uint64_t src_bytes);
/* AtlasTileAttachmentWithData::start_saving (:77-116): writes "{directory}/{coord}.bin" for every
 * existing tile.  TileAtlas::save_tile_config (:605-612): bincode-2 TC file (tiles sorted). */
bt_status bt_atlas_save_attachment(bt_atlas* atlas, uint32_t attachment_index, const char* directory);
bt_status bt_atlas_save_tile_config(const bt_atlas* atlas, const char* file_path);
/* load_tile_config (:616-623): marks the listed tiles as existing. */
bt_status bt_atlas_load_tile_config(bt_atlas* atlas, const char* file_path);
/* formats/mod.rs:8-35 — bincode 2 `config::standard()` of Vec<TileCoordinate>. */
uint64_t bt_tc_encode(const bt_tile_coordinate* tiles, uint32_t count, uint8_t* out, uint64_t cap);
int64_t bt_tc_decode(const uint8_t* data, uint64_t bytes, bt_tile_coordinate* tiles, uint32_t cap);

/* AttachmentData::generate_mipmaps (terrain_data/mod.rs:143-219) on the GPU.
 * Single tile, host in/out: `out` receives all levels concatenated (level 0 first). */
bt_status bt_generate_mipmaps(bt_ctx* ctx, uint32_t format, uint32_t texture_size, uint32_t mip_level_count,
                              const void* level0_host, void* out_host, uint64_t out_bytes);
/* Whole atlas: builds mip levels 1.. of `count` layers starting at `first_layer` into the atlas's mip
 * storage (GpuAtlasAttachment::new allocates mip_level_count levels, gpu_tile_atlas.rs:195-237). */
bt_status bt_atlas_generate_mipmaps(bt_atlas* atlas, uint32_t attachment_index, uint32_t first_layer, uint32_t count);
/* The tile load path — AtlasTileAttachmentWithData::start_loading (tile_atlas.rs:118-149) + upload_tiles
 * (gpu_tile_atlas.rs:309-336) for a batch: reads "{directory}/{coord}.bin" of each tile into its atlas layer
 * (allocated on demand) and builds the mip levels of those layers on the GPU.  coords == NULL: every tile
 * that load_tile_config marked as existing.  A missing or wrongly sized file is BT_ERR_IO. */
bt_status bt_atlas_load_tiles(bt_atlas* atlas, uint32_t attachment_index, const char* directory,
                              const bt_tile_coordinate* coords, uint32_t count);
bt_status bt_atlas_mip_storage(const bt_atlas* atlas, uint32_t attachment_index, uint32_t mip_level,
                               void** device_ptr, uint64_t* tile_bytes);

/* TileLookup (terrain_data/tile_tree.rs:67-81): the best loaded tile for a position and the uv inside its centre. */
typedef struct bt_tile_lookup {
    uint32_t atlas_index; /* BT_INVALID_ATLAS_INDEX: nothing loaded -> the sample is vec4(0) (tile_atlas.rs:250-252) */
    uint32_t atlas_lod;
    float atlas_uv[2];
} bt_tile_lookup;
/* TileAtlas::sample_attachment -> AtlasAttachment::sample + AttachmentData::sample (tile_atlas.rs:249-258, 569-571;
 * terrain_data/mod.rs:220-263) for a batch of lookups: bilinear sample of the tile's level 0, vec4 per lookup
 * (R16: x = height in [0, 1]).  The reference keeps a CPU copy of every loaded tile for this; here the tiles live in
 * HBM, so the query runs there: `lookups_host` in, `out_vec4_host` (4 floats per lookup) out, synchronous. */
bt_status bt_atlas_sample(bt_atlas* atlas, uint32_t attachment_index, const bt_tile_lookup* lookups_host, uint32_t count,
                          float* out_vec4_host);

/* ------------------------------- Preprocessor (preprocess/preprocessor.rs) */
bt_status bt_preprocessor_create(bt_ctx* ctx, bt_preprocessor** out); /* Preprocessor::new :224-232 */
void bt_preprocessor_destroy(bt_preprocessor* p);
/* clear_attachment (:290-296): existing_tiles.clear(); if `directory` is non-NULL also reset_directory
 * (:18-22): remove "{directory}/../../config.tc", rm -r + mkdir -p the directory. */
bt_status bt_preprocessor_clear_attachment(bt_preprocessor* p, bt_atlas* atlas, uint32_t attachment_index,
                                           const char* directory);
/* preprocess_tile (:298-312) / preprocess_spherical (:314-343): build the task queue and assign atlas
 * indices in the reference's order. Nothing runs on the GPU yet. */
bt_status bt_preprocessor_preprocess_tile(bt_preprocessor* p, bt_atlas* atlas, const bt_preprocess_dataset* dataset,
                                          const bt_raster* source);
bt_status bt_preprocessor_preprocess_spherical(bt_preprocessor* p, bt_atlas* atlas, const bt_spherical_dataset* dataset,
                                               const bt_raster sources[6]);
/* queued task counts in the order split, stitch, downsample, save, barrier (PreprocessTaskType :68-82) */
uint32_t bt_preprocessor_task_counts(const bt_preprocessor* p, uint32_t counts[5]);

enum {
    BT_RUN_AUTO = 0,    /* fused split+pyramid kernels where the job qualifies, else reference-shaped */
    BT_RUN_GENERIC = 1, /* one batched launch per queue phase: split / downsample / stitch */
    BT_RUN_KEEP_QUEUE = 2, /* do not clear the queue (benchmarks re-run the same queue) */
    BT_RUN_PROFILE = 4,    /* record a hipEvent after every launch; read with bt_preprocessor_profile() */
    BT_RUN_SHARD_LOCAL = 8,   /* sharded run, part 1: this rank's column strip of the finest LODs (no collective) */
    BT_RUN_SHARD_FINISH = 16, /* sharded run, part 2 (after the all-gathers): cross-strip aprons + the top LODs */
    BT_RUN_SHARD_DISTRIBUTED = 32, /* sharded planar job: the finest LOD is NOT exchanged — its tiles stay on the rank that
                                    * computed them (they are complete there: finest aprons come from the source), only the
                                    * two parent LODs travel (a quarter of the bytes), every rank still ends with every
                                    * lower LOD.  bt_preprocessor_save then writes this rank's share only.  Jobs with more
                                    * than one side (cube seams read neighbour tiles of the finest LOD): BT_ERR_UNSUPPORTED. */
    BT_RUN_SHARD_EXCHANGE = 64,    /* bt_preprocessor_run_sharded only: issue just the grouped collective of the compiled plan
                                    * (no kernels) — lets a benchmark time the exchange alone */
    BT_RUN_SHARD_OVERLAP = 128,    /* bt_preprocessor_run_sharded only (with BT_RUN_KEEP_QUEUE): the local phase on the context's
                                    * stream, the grouped collective behind it on the communicator's OWN stream, and return — the
                                    * finishing kernels come with bt_preprocessor_finish_sharded.  In between the caller runs the
                                    * local phase of another job (its own preprocessor + atlas): the collective of step k hides
                                    * behind the kernels of step k + 1 */
    BT_RUN_REFERENCE_DISPATCH = 256, /* bt_preprocessor_run / _run_streamed: reproduce the reference's dispatch for texture sizes
                                    * that are not multiples of 8 — it launches texture_size / 8 workgroup rows of 8 x 8
                                    * (src/terrain_data/gpu_tile_atlas.rs:105), so split, downsample and stitch never write the
                                    * last texture_size % 8 rows of a tile (they keep what the atlas held).  Default: every row is
                                    * processed.  No effect on attachments whose texture size is a multiple of 8 */
};
/* Replaces select_ready_tasks + GpuPreprocessor::prepare + TerrainPreprocessNode::run for the whole
 * queue: enqueues every kernel on the context's stream and returns (asynchronous).  Save tasks are
 * remembered; bt_preprocessor_save() writes their files. */
bt_status bt_preprocessor_run(bt_preprocessor* p, bt_atlas* atlas, uint32_t flags);
/* Executes the pending Save tasks: "{assets_root}/{config.path}/data/{name}/{coord}.bin" and, like
 * select_ready_tasks on completion (:358-371), "{assets_root}/{config.path}/config.tc". */
bt_status bt_preprocessor_save(bt_preprocessor* p, bt_atlas* atlas, const char* assets_root);
/* The reference's whole span (preprocessor.rs:363,419: all sources loaded -> all saves done) as one overlapped pipeline:
 * = bt_preprocessor_run + bt_preprocessor_save, but every fused main / direct launch of the plan whose sources are
 * BT_RASTER_HOST_DEFERRED rasters runs in bands of tile rows: a band's source rows travel to the GPU, its kernels start when they
 * have landed, and its finished tiles are downloaded and written while later bands (of the same raster, of the next cube face, of the
 * next attachment) upload and run — three HIP queues, one extra host thread.  Since ABI 6 that covers both of the reference's examples:
 * several attachments in one queue (examples/preprocess_planar.rs:16-60: height R16 + albedo Rgba8, one dataset each) and the six face
 * rasters of a cube job (examples/preprocess_spherical.rs:20-48; the finest tiles on a face edge wait for the seam stitch at the end,
 * all others leave with their band).  Launches that cannot be banded (generic plan, device rasters) run whole, in plan order; a queue
 * without any bandable launch runs the legs one after the other.  Files are byte-identical either way.
 * Synchronous: returns when every file is written.  flags: BT_RUN_GENERIC / BT_RUN_KEEP_QUEUE / BT_RUN_REFERENCE_DISPATCH. */
typedef struct bt_stream_stats {
    uint32_t streamed; /* 1: the overlapped pipeline ran; 0: upload, kernels and save ran one after the other */
    uint32_t bands;    /* bands of all banded launches together */
    uint32_t banded_launches; /* fused main / direct launches that ran band by band (ABI 6) */
    uint32_t early_tiles;     /* tiles downloaded and written before the last launch was issued (ABI 6) */
    uint64_t uploaded_bytes;  /* source bytes that travelled in this call (a sharded rank: its windows only) (ABI 6) */
    uint64_t saved_bytes;     /* tile bytes this call downloaded and wrote (a sharded rank: its share) (ABI 6) */
} bt_stream_stats;
bt_status bt_preprocessor_run_streamed(bt_preprocessor* p, bt_atlas* atlas, const char* assets_root, uint32_t flags, bt_stream_stats* out);
/* Launch statistics of the last bt_preprocessor_run: kernels launched, algorithmic bytes
 * (source texels read once + tile texels written once), tiles produced. */
typedef struct bt_run_stats {
    uint32_t kernel_launches;
    uint32_t tiles;
    uint64_t algorithmic_bytes;
    uint32_t fused_jobs, generic_jobs;
    uint32_t prev_zero_launches; /* (ABI 6) fused main / direct launches of the LAST run whose finest tiles nothing had written since
                                  * bt_atlas_create: "the previous value" of a no-data pixel (split.wgsl:34-42) was taken as the
                                  * atlas's initial 0 instead of fetched — same bytes, no atlas reads.  0 for re-runs of a kept queue */
    uint32_t reserved;
} bt_run_stats;
bt_status bt_preprocessor_last_run_stats(const bt_preprocessor* p, bt_run_stats* out);
/* Multi-GPU: tiles shard by column strips of the finest LODs (new design, the reference is single-GPU;
 * SURVEY.md §8e).  Every rank builds the SAME queue (same atlas indices), calls set_shard(rank, world),
 * runs BT_RUN_SHARD_LOCAL, all-gathers each returned range IN PLACE over its atlas storage
 * (ncclAllGather with sendbuff = recvbuff + rank * count: layers [first_layer + r * layers_per_rank, ...) hold
 * rank r's tiles because atlas indices are x-major), then runs BT_RUN_SHARD_FINISH.  Every rank ends with the
 * full atlas, bit-identical to a single-GPU run (with BT_RUN_SHARD_DISTRIBUTED: with its own finest tiles and every
 * lower LOD — the ranges / pieces of the finest LOD are then skipped).  world == 1 restores the normal behaviour. */
typedef struct bt_shard_range {
    uint32_t attachment_index, side, lod;
    uint32_t first_layer;     /* atlas index of tile (x = 0, y = 0) of this (side, lod) */
    uint32_t layers_per_rank; /* contiguous layers owned by each rank */
} bt_shard_range;
bt_status bt_preprocessor_set_shard(bt_preprocessor* p, uint32_t rank, uint32_t world);
/* valid after the first run of the current queue; *count = 0 means "not sharded: every rank computed everything"
 * (or: sharded, but not the regular one-side layout — see bt_preprocessor_shard_pieces) */
/* The window [x0, x1) x [y0, y1) (window = {x0, y0, x1, y1}) of source raster `raster_index` (rasters in the order of the
 * preprocess_* calls; preprocess_spherical adds six) that this preprocessor's launches read; compiles the plan if necessary
 * (flags: BT_RUN_GENERIC).  For a sharded preprocessor with a fused plan that is this rank's column strips + halo — a rank needs
 * no other texel of the source: a BT_RASTER_HOST_DEFERRED raster is uploaded window-only by bt_preprocessor_run, and a caller
 * that fills a device raster itself fills only this (bench.py --gpus N generates only the window).  Otherwise the whole raster.
 * uploaded_bytes (may be NULL): the bytes of the last deferred raster that actually travelled. */
bt_status bt_preprocessor_source_window(bt_preprocessor* p, bt_atlas* atlas, uint32_t raster_index, uint32_t flags, uint32_t window[4],
                                        uint64_t* uploaded_bytes);
bt_status bt_preprocessor_shard_ranges(const bt_preprocessor* p, bt_shard_range* out, uint32_t cap, uint32_t* count);
/* The general form of the exchange (planar AND cube jobs): ownership goes by UNITS — one column strip of one cube side
 * at the granularity of the coarsest LOD the main kernel produces, numbered side-major, `units / world` consecutive
 * units per rank (a job shards iff world divides the unit count: 16k planar 8 units, the 6-face cube job 24).  Every
 * piece is a run of consecutive atlas layers computed by `owner_rank` alone; after BT_RUN_SHARD_LOCAL each piece is
 * broadcast in place from its owner (all pieces inside ONE ncclGroupStart / ncclGroupEnd), then BT_RUN_SHARD_FINISH. */
typedef struct bt_shard_piece {
    uint32_t attachment_index, side, lod;
    uint32_t first_layer, layers;
    uint32_t owner_rank;
} bt_shard_piece;
bt_status bt_preprocessor_shard_pieces(const bt_preprocessor* p, bt_shard_piece* out, uint32_t cap, uint32_t* count);

/* RCCL communicator of the sharded path (new design; `backend "nccl"` IS RCCL on ROCm).  The library resolves the RCCL
 * entry points at run time (the librccl already in the process, else librccl.so.1) — no link-time dependency, and a
 * host that owns an ncclComm_t can hand it over with bt_comm_adopt. */
#define BT_COMM_UNIQUE_ID_BYTES 128
typedef struct bt_comm bt_comm;
bt_status bt_comm_unique_id(uint8_t out[BT_COMM_UNIQUE_ID_BYTES]); /* ncclGetUniqueId: call on rank 0, ship to the others */
bt_status bt_comm_create(bt_ctx* ctx, uint32_t world, uint32_t rank, const uint8_t unique_id[BT_COMM_UNIQUE_ID_BYTES], bt_comm** out);
bt_status bt_comm_adopt(bt_ctx* ctx, void* nccl_comm, uint32_t world, uint32_t rank, bt_comm** out); /* borrowed ncclComm_t */
void bt_comm_destroy(bt_comm* comm);
/* health check: a small grouped in-place all-gather + broadcast through the communicator, verified on the host */
bt_status bt_comm_check(bt_comm* comm);
/* The same with slots of `slot_bytes` each (one atlas tile, say): ONE grouped collective — an in-place all-gather of one slot per rank
 * and an in-place broadcast from the last rank — on the context's stream, every byte verified on the host.  Meant to run once before the
 * first sharded step: a communicator that cannot move a tile fails HERE with RCCL's error string (or, if the collective hangs, under
 * the caller's watchdog) instead of inside a timed step.  `elapsed_ms` (may be NULL): device time of the collective.
 * slot_bytes: 1 .. 64 MiB (world + 1 slots are allocated on the device and on the host).  A HANG IS NOT DETECTED: when the collective
 * fails on some ranks only, the others block in the stream synchronisation — run it under a watchdog (bench.py --preflight-timeout). */
bt_status bt_comm_preflight(bt_comm* comm, uint64_t slot_bytes, float* elapsed_ms);
/* One step of a sharded job, entirely on the context's stream and without host synchronisation: this rank's strip
 * (BT_RUN_SHARD_LOCAL), ONE grouped collective (in-place ncclAllGather per LOD for the regular planar layout, in-place
 * ncclBroadcast per piece otherwise, between ncclGroupStart and ncclGroupEnd), the finishing kernels
 * (BT_RUN_SHARD_FINISH).  `flags`: BT_RUN_GENERIC / BT_RUN_KEEP_QUEUE / BT_RUN_PROFILE, and BT_RUN_SHARD_LOCAL alone to
 * skip the collective and the finish (kernel-only timing).  set_shard(rank, world) must match the communicator. */
bt_status bt_preprocessor_run_sharded(bt_preprocessor* p, bt_atlas* atlas, bt_comm* comm, uint32_t flags);
/* Second half of a BT_RUN_SHARD_OVERLAP step: the context's stream waits for that step's collective, then the finishing kernels
 * run.  flags: BT_RUN_GENERIC / BT_RUN_SHARD_DISTRIBUTED as in the first half, BT_RUN_PROFILE, BT_RUN_KEEP_QUEUE. */
bt_status bt_preprocessor_finish_sharded(bt_preprocessor* p, bt_atlas* atlas, bt_comm* comm, uint32_t flags);
/* The end-to-end span of a SHARDED job with a distributed result (BT_RUN_SHARD_DISTRIBUTED is implied; planar jobs): every rank is one
 * PCIe link.  Rank r uploads only its source window band by band (bt_preprocessor_source_window), runs its units, writes its finest
 * tiles band by band while later bands upload and run, exchanges the two parent LODs (the grouped collective of
 * bt_preprocessor_run_sharded, on the context's stream), runs the finishing kernels and writes its share of the lower LODs (every
 * world-th tile; config.tc from rank 0): the ranks together produce the reference's directory, one writer per file.
 * `comm` may be NULL for hosts that bring their own collective: call once with BT_RUN_SHARD_LOCAL (upload + local kernels + this rank's
 * finest files), exchange bt_preprocessor_shard_ranges / _pieces below the finest LOD yourself, call again with BT_RUN_SHARD_FINISH
 * (finishing kernels + the rest of this rank's files).  With `comm`: pass both flags (or neither) — one call does everything.
 * Other flags: BT_RUN_KEEP_QUEUE.  A preprocessor that is not sharded (world 1) behaves like bt_preprocessor_run_streamed. */
bt_status bt_preprocessor_run_streamed_sharded(bt_preprocessor* p, bt_atlas* atlas, bt_comm* comm, const char* assets_root, uint32_t flags,
                                               bt_stream_stats* out);

/* Per-launch device time of the runs made with BT_RUN_PROFILE since the last call (hipEvents on the
 * context's stream, averaged over those runs).  `kind`: 0 split, 1 downsample, 2 stitch, 3 fused main,
 * 4 fused tail, 5 fused direct (Rgba8), 6 fused todo (re-queued no-data chunks + apron corners, follows 3).  `algorithmic_bytes`: that launch's inputs read once + outputs written once.
 * Synchronises the stream.  Returns the number of launches per run through *count. */
typedef struct bt_launch_profile {
    uint32_t kind;
    uint32_t tasks;
    uint64_t algorithmic_bytes;
    float avg_ms;
    uint32_t samples;
} bt_launch_profile;
bt_status bt_preprocessor_profile(bt_preprocessor* p, bt_launch_profile* out, uint32_t cap, uint32_t* count);

/* ---------------- tiling prepass (render/tiling_prepass.rs, shaders/tiling_prepass) */
/* SideParameter fields the prepass reads (math/terrain_model.rs:228-233; named view_xy/view_uv in
 * types.wgsl:78-80). */
typedef struct bt_side_parameter {
    int32_t view_xy[2];
    float view_uv[2];
} bt_side_parameter;

/* Everything `refine_tiles` reads each frame. */
typedef struct bt_view_state {
    uint32_t spherical;                  /* SPHERICAL shader def (tiling_prepass.rs:61-78) */
    uint32_t geometry_tile_count;        /* TerrainViewConfigUniform (terrain_view_bind_group.rs:81-116) */
    uint32_t refinement_count;
    uint32_t vertices_per_tile;
    float subdivision_distance;
    uint32_t origin_lod;                 /* TerrainModelApproximation (terrain_model.rs:252-259) */
    float approximate_height;
    bt_side_parameter sides[6];
    float world_position[3];             /* CullingUniform.world_position (culling_bind_group.rs:41-55) */
    float world_from_local[12];          /* mesh[0].world_from_local: 3x3 columns then translation */
    float local_from_world_transpose[9]; /* mesh[0].local_from_world_transpose_{a,b}: 3x3 columns */
} bt_view_state;

/* Indirect (terrain_view_bind_group.rs:65-71) as prepare_render leaves it. */
typedef struct bt_indirect {
    uint32_t vertex_count, instance_count, base_vertex, base_instance;
} bt_indirect;

/* TerrainViewData::new (:130-142): final_tiles + temporary_tiles of `geometry_tile_count` entries. */
bt_status bt_tiling_prepass_create(bt_ctx* ctx, uint32_t geometry_tile_count, bt_tiling_prepass** out);
void bt_tiling_prepass_destroy(bt_tiling_prepass* t);
/* TilingPrepassNode::run (:204-272): prepare_root, refinement_count x (refine_tiles, prepare_next),
 * refine_tiles, prepare_render — as one persistent launch behind one chip-wide launch that precomputes the divide tests
 * (see bt_tiling_prepass_run_plain).  Asynchronous on the context's stream. */
bt_status bt_tiling_prepass_run(bt_tiling_prepass* t, const bt_view_state* view);
/* bt_tiling_prepass_run is two launches: every divide test that can matter is evaluated up front, chip-wide (a window of
 * tiles around the view at every LOD and side; the test depends on tile and view only), then the ordered schedule runs over
 * those bits out of LDS.  This is the single-launch form that evaluates each test inside its pass: the same list in the
 * same order, ~3x the worst-case latency; the checker of the two-launch form. */
bt_status bt_tiling_prepass_run_plain(bt_tiling_prepass* t, const bt_view_state* view);
/* The reference's contract is the SET of final tiles: refine_tiles appends them in the arrival order of a global atomic
 * (shaders/tiling_prepass/refine_tiles.wgsl:13-15, 41).  This form produces that set (and the same indirect arguments and
 * overflow verdict) in arrival order too, without the chain of passes: after the chip-wide divide tests a second chip-wide
 * launch decides every tile from its ancestors' bits.  Flat ~10 us per frame whatever the view; temporary_tiles is not
 * written.  The two entries above additionally reproduce the order of a run with invocations taken in id order. */
bt_status bt_tiling_prepass_run_unordered(bt_tiling_prepass* t, const bt_view_state* view);
/* Window radius (tiles around the view's tile at every LOD that get their divide test up front) of the unordered form:
 * 1..28, 0 = default (28).  A tuning / test knob: results do not depend on it (tiles outside the windows are evaluated
 * in place). */
bt_status bt_tiling_prepass_set_window(bt_tiling_prepass* t, uint32_t radius);
/* Device buffers a renderer binds: final_tiles (bt_tile_coordinate[]), indirect args, counters. */
bt_status bt_tiling_prepass_buffers(const bt_tiling_prepass* t, void** final_tiles_device, void** indirect_device);
/* Synchronises and copies the final tile list (bt_tiling_prepass_run / _run_plain: in the reference's sequential append
 * order; _run_unordered: arrival order). */
bt_status bt_tiling_prepass_read(bt_tiling_prepass* t, bt_tile_coordinate* final_tiles_host, uint32_t cap,
                                 uint32_t* count, bt_indirect* indirect);

/* -------------------------- TerrainModel / TerrainViewConfig / TileTree (the per-frame CPU side of the prepass) */
enum { BT_MODEL_PLANAR = 0, BT_MODEL_SPHERICAL = 1, BT_MODEL_ELLIPSOIDAL = 2 };
/* TerrainModel (math/terrain_model.rs:41-115): rotation is the identity, as in all three reference constructors.
 * planar: a = side_length; sphere: a = radius; ellipsoid: a = major_axis, b = minor_axis (scale = (a, b, a)). */
typedef struct bt_terrain_model {
    uint32_t kind; /* BT_MODEL_* */
    uint32_t _padding;
    double position[3];
    double a, b;
    float min_height, max_height;
} bt_terrain_model;
/* TerrainViewConfig (terrain_view.rs:18-63), same fields and defaults. */
typedef struct bt_terrain_view_config {
    uint32_t tree_size;           /* 8 */
    uint32_t geometry_tile_count; /* 1000000 */
    uint32_t refinement_count;    /* 30 */
    uint32_t grid_size;           /* 16 */
    double subdivision_tolerance; /* 0.1 */
    double precision_threshold_distance; /* 0.001 */
    double load_distance;         /* 2.5 */
    double morph_distance;        /* 16.0 */
    double blend_distance;        /* 2.0 */
    float morph_range;            /* 0.2 */
    float blend_range;            /* 0.2 */
    uint32_t origin_lod;          /* 10 */
    uint32_t _padding;
} bt_terrain_view_config;
void bt_terrain_view_config_default(bt_terrain_view_config* out);

/* Everything the tiling prepass reads for one view and frame, derived the way the reference derives it:
 * TileTree::new (tile_tree.rs:135-173), TerrainViewConfigUniform::from_tile_tree (terrain_view_bind_group.rs:98-116),
 * TerrainModelApproximation::compute (terrain_model.rs:262-290: only origin_xy / origin_uv per side are read by
 * refine_tiles, HIGH_PRECISION is never defined for it), CullingUniform.world_position, and the mesh uniform of
 * TerrainModel::transform() (terrain_model.rs:195-201).  f64 on the host, `as f32` where the reference casts. */
bt_status bt_view_state_from_config(const bt_terrain_model* model, const bt_terrain_view_config* view_config,
                                    const double view_world_position[3], float approximate_height, bt_view_state* out);

typedef struct bt_tile_tree bt_tile_tree; /* TileTree + GpuTileTree of one (terrain, view) pair */
/* TileTree::new (tile_tree.rs:135-173).  The node tables (tile states, TileTreeEntry data, origins) live in HBM. */
bt_status bt_tile_tree_create(bt_ctx* ctx, const bt_terrain_model* model, uint32_t lod_count,
                              const bt_terrain_view_config* view_config, bt_tile_tree** out);
void bt_tile_tree_destroy(bt_tile_tree* tree);
/* TileTree::compute_requests -> update (tile_tree.rs:268-359) as ONE launch over sides x lods x tree_size^2 nodes
 * (f64, like the reference): origins, per-node tile coordinate / distance / request state, and the released and
 * requested tile lists in the reference's push order (stable ballot / prefix-sum compaction).  Synchronises; the lists
 * stay readable until the next update. */
bt_status bt_tile_tree_update(bt_tile_tree* tree, const double view_world_position[3]);
bt_status bt_tile_tree_requests(const bt_tile_tree* tree, const bt_tile_coordinate** released, uint32_t* released_count,
                                const bt_tile_coordinate** requested, uint32_t* requested_count);
/* TileAtlas::update's second half (tile_atlas.rs:590-600): drains the lists into release_tile / request_tile. */
bt_status bt_tile_tree_apply_requests(bt_tile_tree* tree, bt_atlas* atlas);
/* TileTree::adjust_to_tile_atlas (:363-374): every node's TileTreeEntry = get_best_tile(node coordinate), looked up on
 * the GPU in a device copy of the atlas's tile states (refreshed when they changed).  Asynchronous. */
bt_status bt_tile_tree_adjust_to_tile_atlas(bt_tile_tree* tree, const bt_atlas* atlas);
/* GpuTileTree buffers (gpu_tile_tree.rs:22-95), kept current on the device — no per-frame upload:
 * entries = bt_tile_tree_entry[side][lod][x][y], origins = uint32[side][lod][2]. */
bt_status bt_tile_tree_buffers(const bt_tile_tree* tree, void** entries_device, void** origins_device);
/* host copies for inspection / tests (synchronise) */
bt_status bt_tile_tree_read(bt_tile_tree* tree, bt_tile_tree_entry* entries, uint32_t entry_cap, uint32_t* origins_xy,
                            uint32_t origin_cap, bt_tile_coordinate* node_coordinates, uint32_t* node_requested);
/* sample_height / sample_attachment (terrain_data/mod.rs:265-307) for a batch of world positions: surface projection,
 * TileTree::compute_blend (:223-239), lookup_tile (:241-266) at lod and lod - 1, bilinear tile samples
 * (AtlasAttachment::sample) and their blend, on the GPU against the tree's entries and the atlas in HBM.
 * out_vec4: 4 floats per position (the attachment value); heights (optional): lerp(min_height, max_height, value.x). */
bt_status bt_tile_tree_sample_attachment(bt_tile_tree* tree, bt_atlas* atlas, uint32_t attachment_index,
                                         const double* world_positions_xyz, uint32_t count, float* out_vec4, float* heights);
/* TileTree::approximate_height (:376-386): sample_height at the view position of the last update; also kept by the tree
 * for the next update's tile distances.  */
bt_status bt_tile_tree_approximate_height(bt_tile_tree* tree, bt_atlas* atlas, float* height);
/* the tree's current state as a prepass input (bt_view_state_from_config with the tree's view position / height) */
bt_status bt_tile_tree_view_state(const bt_tile_tree* tree, bt_view_state* out);

/* One frame of one view (src/plugin.rs:46-56: TileTree::compute_requests -> TileAtlas::update's release / request half ->
 * TileTree::adjust_to_tile_atlas -> TileTree::approximate_height -> TilingPrepassNode::run) as ONE call with ONE host
 * synchronisation — the one the structure forces: the atlas's streaming state machine is host code and needs the two lists,
 * which the update kernel writes straight into pinned memory.  Everything behind it is enqueued and left running on the
 * context's stream: tile states from pinned staging, the sampled height stays on the device (the prepass kernels and the
 * next frame's update read it there; bt_tile_tree_view_state / the next bt_frame_info report it one frame late), the final
 * tile list and indirect arguments land in bt_tiling_prepass_buffers().  File loads stay with bt_atlas_update().
 * Identical lists, entries and final tiles to the separate calls (tests/test_gpu_tile_tree.py). */
enum {
    BT_FRAME_PREPASS_UNORDERED = 1, /* bt_tiling_prepass_run_unordered instead of bt_tiling_prepass_run */
    BT_FRAME_PREPASS_PLAIN = 2,     /* bt_tiling_prepass_run_plain */
    BT_FRAME_KEEP_REQUESTS = 4,     /* do not apply the lists to the atlas: the caller does (bt_tile_tree_requests / _apply_requests) */
    BT_FRAME_KEEP_HEIGHT = 8,       /* skip approximate_height */
};
typedef struct bt_frame_info {
    uint32_t released_count, requested_count; /* this frame's lists (bt_tile_tree_requests) */
    bt_status apply_status;                   /* of bt_tile_tree_apply_requests (BT_OK when skipped) */
    float approximate_height;                 /* the height this frame's update used, i.e. the previous frame's sample */
} bt_frame_info;
bt_status bt_frame_update(bt_tile_tree* tree, bt_atlas* atlas, bt_tiling_prepass* prepass /* may be NULL */,
                          const double view_world_position[3], uint32_t flags, bt_frame_info* out /* may be NULL */);

/* ---------------------------------------------------------------- diagnostics */
/* Exhaustive device check that the kernels' 3-operation unorm16 -> f32 conversion equals the correctly
 * rounded division t / 65535.0f for all 65536 texel values; *failures must come back 0. */
bt_status bt_selftest(bt_ctx* ctx, uint32_t* failures);

/* ---------------------------------------------------------------- synthetic inputs */
/* Deterministic integer fBm heightmap in [1, 65535] written straight into HBM (bench / tests: the
 * reference's Gaia and GEBCO source rasters are not in its checkout).  The window
 * (x0, y0, width, height) of the infinite field is produced, so shards can generate only their part. */
bt_status bt_synth_fbm_r16(bt_ctx* ctx, void* dst_device, uint32_t width, uint32_t height, uint64_t row_pitch,
                           uint32_t x0, uint32_t y0, uint32_t base_cell, uint32_t octaves, uint32_t seed);

#ifdef __cplusplus
}
#endif
#endif /* BEVY_TERRAIN_AMD_H */
