/*
 * bt_oracle.h — CPU ORACLE.  TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C restatement of the reference's terrain-tile preprocessing and
 * tiling-prepass algorithms (kurtkuehnert/bevy_terrain @ 2025-03-14), used
 * only as the checker in tests/, __graft_entry__.smoke() and the
 * `cpu_baseline` leg of bench.py.  Nothing in the product path
 * (bevy_terrain_amd/, include/) may include, link or call this.
 *
 * PARITY: pinned to the reference's own shader text since round 3.  The reference ships no tests, golden vectors or
 * fixtures and its Rust / wgpu host cannot be built here; but the arithmetic of the path is WGSL
 * text, and oracle/wgsl_ref translates that text mechanically (wgsl2cpp.py) and executes it (oracle/_ref): every Split /
 * Downsample / Stitch task and the whole tiling prepass of this file are checked bit for bit against it
 * (tests/test_wgsl_ref.py; orc_set_task_backend below is the seam), and tests/golden/ are its outputs.  What this file
 * restates from RUST code (queue order and atlas indices, coordinate.rs, the f64 tile tree / terrain model in
 * bt_oracle_tree.c, generate_mipmaps, bincode) has no executable counterpart and stays restated, statement by statement
 * with the file:line it follows.  Where the reference delegates to the GPU driver (hardware bilinear sampler, unorm
 * conversions, 0/0) the definition chosen is written next to the function and is the same in oracle/wgsl_ref.
 */
#ifndef BT_ORACLE_H
#define BT_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INVALID 0xFFFFFFFFu /* terrain_data/mod.rs:34 INVALID_ATLAS_INDEX */

/* AttachmentFormat::id — terrain_data/mod.rs:50-57 */
#define ORC_FORMAT_RGBA8 0u
#define ORC_FORMAT_R16 1u

/* math/coordinate.rs:155-167 */
typedef struct {
    uint32_t side, lod, x, y;
} orc_coord;

/* terrain_data/tile_atlas.rs:30-35 (32 bytes with padding) */
typedef struct {
    orc_coord coordinate;
    uint32_t atlas_index;
    uint32_t _pad[3];
} orc_atlas_tile;

/* terrain_data/mod.rs:87-109 */
typedef struct {
    uint32_t texture_size;
    uint32_t border_size;
    uint32_t mip_level_count;
    uint32_t format; /* ORC_FORMAT_* */
} orc_attachment_config;

/* preprocess/preprocessor.rs:35-55 */
typedef struct {
    uint32_t attachment_index;
    uint32_t side;
    float top_left[2];
    float bottom_right[2];
    uint32_t lod_begin, lod_end; /* lod_range = lod_begin..lod_end */
} orc_dataset;

typedef struct orc_atlas orc_atlas;

/* ---- math/coordinate.rs ------------------------------------------------ */
void orc_children(orc_coord c, orc_coord out[4]);
void orc_neighbours(orc_coord c, int spherical, orc_coord out[8]);
orc_coord orc_parent(orc_coord c);
int orc_coord_is_invalid(orc_coord c);
/* "{side}_{lod}_{x}_{y}" — coordinate.rs:282-286 */
int orc_coord_name(orc_coord c, char* buf, size_t n);

/* ---- terrain_data/tile_atlas.rs + preprocess ---------------------------- */
orc_atlas* orc_atlas_new(uint32_t lod_count, uint32_t atlas_size, int spherical,
                         uint32_t n_attachments, const orc_attachment_config* att);
void orc_atlas_free(orc_atlas* a);
void orc_clear_attachment(orc_atlas* a, uint32_t attachment_index);
/* queue construction (preprocessor.rs:234-343); src rasters are W x H, tightly
 * packed u16 (R16) or RGBA8, kept by pointer until orc_run(). */
int orc_preprocess_tile(orc_atlas* a, const orc_dataset* d, const void* src, uint32_t w, uint32_t h);
int orc_preprocess_spherical(orc_atlas* a, uint32_t attachment_index, uint32_t lod_begin,
                             uint32_t lod_end, const void* const src[6], uint32_t w, uint32_t h);
/* executes the queued tasks in order (barriers = sequential phases);
 * threads<=1 -> scalar, else OpenMP over the tasks between two barriers. */
int orc_run(orc_atlas* a, int threads);
uint32_t orc_task_count(const orc_atlas* a, uint32_t counts[5]);
/* The bench's CPU-baseline schedule of the same queue (identical results): units = (task, block of rows), in-place stores,
 * wall time and task count per phase.  orc_atlas_touch first-touches the atlas pages outside the timed region. */
int orc_run_blocks(orc_atlas* a, int threads, uint32_t rows_per_block, double* phase_seconds, uint32_t* phase_tasks, uint32_t cap,
                   uint32_t* n_phases);
void orc_atlas_touch(orc_atlas* a, int threads);

uint32_t orc_tile_count(const orc_atlas* a); /* existing_tiles.len() */
/* existing tiles in atlas-index order + their atlas indices */
uint32_t orc_tiles(const orc_atlas* a, orc_coord* coords, uint32_t* atlas_indices, uint32_t cap);
uint32_t orc_get_tile(const orc_atlas* a, orc_coord c); /* atlas index or INVALID */
const void* orc_tile_data(const orc_atlas* a, uint32_t attachment_index, uint32_t atlas_index);
size_t orc_tile_bytes(const orc_atlas* a, uint32_t attachment_index);

/* tile_atlas.rs:77-116 + 605-612 : "{root}/data/{name}/{coord}.bin", "{root}/config.tc" */
int orc_save_attachment(const orc_atlas* a, uint32_t attachment_index, const char* dir);
int orc_save_tile_config(const orc_atlas* a, const char* path);
/* formats/mod.rs:8-35 (bincode 2 standard config) */
size_t orc_tc_encode(const orc_coord* tiles, uint32_t n, uint8_t* out, size_t cap);
long orc_tc_decode(const uint8_t* in, size_t n, orc_coord* tiles, uint32_t cap);

/* ---- task backend: run the queue with someone else's kernels ------------ */
/* The queue driver (orc_run) normally executes each Split / Downsample / Stitch task with this file's restated
 * kernels.  A backend replaces ONLY that step: it receives the task exactly as the reference's GPU path would (tile,
 * related tiles, dataset rectangle, source raster, the attachment's atlas as it stands) and must return the tile's new
 * contents (tight T x T texels) in out_tile; the driver stores it after the task, as the write-section copy-back does.
 * Used by tests to run the queue through oracle/_ref (the reference's WGSL executed on the CPU). */
enum { ORC_TASK_SPLIT = 0, ORC_TASK_STITCH = 1, ORC_TASK_DOWNSAMPLE = 2 };
typedef struct {
    int32_t type;          /* ORC_TASK_* */
    uint32_t format, lod_count, texture_size, border_size, atlas_size;
    orc_atlas_tile tile;
    orc_atlas_tile rel[8]; /* children (4) or neighbours (8) */
    float top_left[2], bottom_right[2];
    const void* src;       /* split: source raster, tightly packed */
    uint32_t src_w, src_h;
    const void* atlas;     /* this attachment's layers: atlas_size x T x T texels, tightly packed */
} orc_task_desc;
typedef void (*orc_task_backend)(void* user, const orc_task_desc* task, void* out_tile);
void orc_set_task_backend(orc_atlas* a, orc_task_backend backend, void* user); /* NULL restores the built-in kernels */

/* Sampler model for split (process-wide; test infrastructure): fractional_bits = 0 -> exact f32 bilinear (the
 * definition the product implements); N > 0 -> filter weights snapped to N fractional bits, mode 0 to nearest,
 * mode 1 truncated — what the hardware sampler behind split.wgsl:32 may do. */
void orc_set_sampler_model(int fractional_bits, int mode);

/* ---- per-task kernels exposed for unit tests --------------------------- */
/* one pixel of split.wgsl:18-43; prev = previous atlas texel (unorm ints, 4 ch) */
void orc_split_pixel(uint32_t format, uint32_t T, uint32_t b, orc_coord tile, const float tl[2],
                     const float br[2], const void* src, uint32_t w, uint32_t h, uint32_t px,
                     uint32_t py, const uint32_t prev[4], uint32_t out[4]);

/* ---- terrain_data/mod.rs:143-219 -------------------------------------- */
/* out holds all levels concatenated (level 0 copied first); returns texels written */
/* AtlasAttachment::sample (tile_atlas.rs:249-258) + AttachmentData::sample (terrain_data/mod.rs:220-263): bilinear
 * sample of level 0 of one tile at `atlas_uv` (the TileLookup's uv over the tile's centre).  out = vec4 (R16: x only).
 * Vec4::lerp is glam's `a + (b - a) * s` (bevy 0.14.0 pins glam 0.27; glam is not vendored in the reference).
 * Defined here where the reference would panic (index out of bounds): texel coordinates are clamped into the tile. */
void orc_sample_tile(uint32_t format, uint32_t texture_size, uint32_t border_size, const void* level0,
                     const float atlas_uv[2], float out[4]);
size_t orc_generate_mipmaps(uint32_t format, uint32_t texture_size, uint32_t mip_level_count,
                            const void* level0, void* out);

/* ---- tiling prepass (shaders/tiling_prepass/*.wgsl, functions.wgsl) ----- */
typedef struct {
    int32_t view_xy[2];
    float view_uv[2];
} orc_side_parameter; /* terrain_model.rs:228-233 (only fields the prepass reads) */

typedef struct {
    uint32_t spherical;            /* SPHERICAL shader def, tiling_prepass.rs:61-78 */
    uint32_t tile_count;           /* view_config.tile_count = geometry_tile_count */
    uint32_t refinement_count;     /* tiling_prepass.rs:251 */
    uint32_t vertices_per_tile;    /* terrain_view_bind_group.rs:106 */
    float subdivision_distance;    /* terrain_view_bind_group.rs:111 */
    uint32_t origin_lod;           /* terrain_model.rs:254 */
    float approximate_height;      /* terrain_model.rs:255 */
    orc_side_parameter sides[6];
    float world_position[3];       /* culling_bind_group.rs:50 */
    float world_from_local[12];    /* mesh[0].world_from_local: 3 columns + translation, column-major */
    float local_from_world_transpose[9]; /* mesh[0].local_from_world_transpose_{a,b}, column-major 3x3 */
} orc_view;

/* runs prepare_root, refinement_count x (refine_tiles, prepare_next), refine_tiles,
 * prepare_render sequentially in invocation-id order.  final_tiles gets the list in
 * append order; returns count, or -1 if the buffers overflowed. indirect[4] = draw args. */
long orc_refine(const orc_view* v, orc_coord* final_tiles, uint32_t cap, uint32_t indirect[4],
                uint32_t* passes_tile_counts /* refinement_count+1 entries or NULL */);
/* refine_tiles.wgsl:17-22 for one tile (exposed for tests) */
int orc_should_be_divided(const orc_view* v, orc_coord tile, float* view_distance);

/* ======================================================================== */
/* TileTree / streaming TileAtlasState / view-state derivation (f64 CPU side) */
/* restated in oracle/bt_oracle_tree.c                                        */
/* ======================================================================== */
/* TerrainModel (math/terrain_model.rs:41-138); identity rotation as in the reference constructors.
 * kind 0 planar (a = side_length), 1 sphere (a = radius), 2 ellipsoid (a = major, b = minor axis). */
typedef struct {
    uint32_t kind, _pad;
    double position[3];
    double a, b;
    float min_height, max_height;
} orc_model;
/* TerrainViewConfig (terrain_view.rs:18-63) */
typedef struct {
    uint32_t tree_size, geometry_tile_count, refinement_count, grid_size;
    double subdivision_tolerance, precision_threshold_distance, load_distance, morph_distance, blend_distance;
    float morph_range, blend_range;
    uint32_t origin_lod, _pad;
} orc_view_config;
typedef struct {
    uint32_t atlas_index, atlas_lod;
} orc_tree_entry; /* tile_tree.rs:49-66 */

/* TerrainViewConfigUniform::from_tile_tree + TerrainModelApproximation::compute (origin_xy / origin_uv) + the mesh
 * uniform: everything refine_tiles reads (terrain_view_bind_group.rs:98-116, terrain_model.rs:262-290, tile_tree.rs:135-173) */
void orc_view_state_from_config(const orc_model* model, const orc_view_config* vc, const double view_world_position[3],
                                float approximate_height, orc_view* out);
/* Coordinate::from_world_position (coordinate.rs:69-113) -> side, uv; world_position (:115-135) */
uint32_t orc_coordinate_from_world_position(const orc_model* model, const double world[3], double uv[2]);
void orc_coordinate_world_position(const orc_model* model, uint32_t side, const double uv[2], float height, double out[3]);
/* math/ellipsoid.rs */
void orc_project_point_ellipsoid(const double e[3], const double y[3], double out[3]);

/* the streaming half of TileAtlasState (tile_atlas.rs:279-503) */
typedef struct orc_stream orc_stream;
orc_stream* orc_stream_new(uint32_t atlas_size, uint32_t attachment_count);
void orc_stream_free(orc_stream* s);
void orc_stream_add_existing(orc_stream* s, const orc_coord* tiles, uint32_t n);
int orc_stream_request_tile(orc_stream* s, orc_coord c);  /* 0, or -2 "Atlas out of indices" */
int orc_stream_release_tile(orc_stream* s, orc_coord c);  /* 0, or -1 "Tried releasing a tile, which is not present." */
orc_tree_entry orc_stream_get_best_tile(const orc_stream* s, orc_coord c);
uint32_t orc_stream_pending_loads(const orc_stream* s);
/* the first `n` queued loads finish (loaded_tile_attachment, :347-359); writes their (coordinate, atlas_index,
 * attachment_index) triples to out (5 x u32 each) when non-NULL; returns how many finished */
uint32_t orc_stream_finish_loads(orc_stream* s, uint32_t n, uint32_t* out);
uint32_t orc_stream_atlas_index(const orc_stream* s, orc_coord c); /* tile_states[c].atlas_index or INVALID */

/* TileTree (tile_tree.rs:103-387) */
typedef struct orc_tile_tree orc_tile_tree;
orc_tile_tree* orc_tile_tree_new(const orc_model* model, uint32_t lod_count, const orc_view_config* vc);
void orc_tile_tree_free(orc_tile_tree* t);
void orc_tile_tree_update(orc_tile_tree* t, const double view_world_position[3]); /* :268-333 */
uint32_t orc_tile_tree_released(const orc_tile_tree* t, orc_coord* out, uint32_t cap);
uint32_t orc_tile_tree_requested(const orc_tile_tree* t, orc_coord* out, uint32_t cap);
int orc_tile_tree_apply_requests(orc_tile_tree* t, orc_stream* s);      /* TileAtlas::update :590-600 */
void orc_tile_tree_adjust_to_tile_atlas(orc_tile_tree* t, const orc_stream* s); /* :363-374 */
uint32_t orc_tile_tree_node_count(const orc_tile_tree* t);
/* entries / node coordinates / node states in [side][lod][x][y] slot order; origins [side][lod][2] */
void orc_tile_tree_read(const orc_tile_tree* t, orc_tree_entry* entries, uint32_t* origins, orc_coord* nodes, uint32_t* requested);
void orc_tile_tree_set_approximate_height(orc_tile_tree* t, float h);
/* compute_blend (:223-239) */
void orc_tile_tree_compute_blend(const orc_tile_tree* t, const double sample_world_position[3], uint32_t* lod, float* ratio);
/* sample_attachment / sample_height (terrain_data/mod.rs:265-307).  layers[atlas_index] = level-0 texels of that slot
 * (NULL where nothing is loaded). */
void orc_tile_tree_sample_attachment(const orc_tile_tree* t, uint32_t format, uint32_t texture_size, uint32_t border_size,
                                     const void* const* layers, uint32_t atlas_size, const double* positions, uint32_t n,
                                     float* out_vec4, float* heights);

#ifdef __cplusplus
}
#endif
#endif
