/* A C99 consumer of include/bevy_terrain_amd.h (what a Rust `extern "C"` block or any other host sees): compiles the
 * header as C, pins the layout of every struct the ABI exchanges with static assertions, prints the layouts as JSON
 * (tests/test_host_logic.py compares them with the ctypes mirror in bevy_terrain_amd/_ffi.py), and drives
 * create / destroy through the shared library loaded with dlopen.
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude tests/abi_consumer.c -ldl -o abi_consumer && ./abi_consumer lib.so */
#include <dlfcn.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>

#include "bevy_terrain_amd.h"

#define STATIC_ASSERT(cond, name) typedef char static_assert_##name[(cond) ? 1 : -1]

STATIC_ASSERT(sizeof(bt_tile_coordinate) == 16, tile_coordinate_is_16_bytes);       /* types.wgsl:25-29 */
STATIC_ASSERT(sizeof(bt_atlas_tile) == 32 && offsetof(bt_atlas_tile, atlas_index) == 16, atlas_tile_is_32_bytes); /* preprocessing.wgsl:16-22 */
STATIC_ASSERT(sizeof(bt_tile_tree_entry) == 8, tile_tree_entry_is_8_bytes);
STATIC_ASSERT(sizeof(bt_attachment_config) == 80 && offsetof(bt_attachment_config, texture_size) == 64, attachment_config);
STATIC_ASSERT(sizeof(bt_terrain_config) == 16 + 8 * 80 + 256 && offsetof(bt_terrain_config, path) == 16 + 8 * 80, terrain_config);
STATIC_ASSERT(sizeof(bt_raster) == 32 && offsetof(bt_raster, row_pitch) == 16 && offsetof(bt_raster, format) == 24, raster);
STATIC_ASSERT(sizeof(bt_preprocess_dataset) == 32 && offsetof(bt_preprocess_dataset, lod_begin) == 24, preprocess_dataset);
STATIC_ASSERT(sizeof(bt_spherical_dataset) == 12, spherical_dataset);
STATIC_ASSERT(sizeof(bt_tile_lookup) == 16, tile_lookup);
STATIC_ASSERT(sizeof(bt_run_stats) == 32 && offsetof(bt_run_stats, algorithmic_bytes) == 8 && offsetof(bt_run_stats, prev_zero_launches) == 24, run_stats);
STATIC_ASSERT(sizeof(bt_stream_stats) == 32 && offsetof(bt_stream_stats, uploaded_bytes) == 16, stream_stats);
STATIC_ASSERT(sizeof(bt_shard_range) == 20, shard_range);
STATIC_ASSERT(sizeof(bt_launch_profile) == 24 && offsetof(bt_launch_profile, avg_ms) == 16, launch_profile);
STATIC_ASSERT(sizeof(bt_side_parameter) == 16, side_parameter);
STATIC_ASSERT(sizeof(bt_view_state) == 28 + 6 * 16 + 12 + 48 + 36 && offsetof(bt_view_state, sides) == 28 &&
                  offsetof(bt_view_state, world_position) == 124 && offsetof(bt_view_state, world_from_local) == 136 &&
                  offsetof(bt_view_state, local_from_world_transpose) == 184,
              view_state);
STATIC_ASSERT(sizeof(bt_indirect) == 16, indirect);
STATIC_ASSERT(sizeof(bt_terrain_model) == 56 && offsetof(bt_terrain_model, position) == 8 && offsetof(bt_terrain_model, a) == 32 &&
                  offsetof(bt_terrain_model, min_height) == 48,
              terrain_model);
STATIC_ASSERT(sizeof(bt_terrain_view_config) == 72 && offsetof(bt_terrain_view_config, subdivision_tolerance) == 16 &&
                  offsetof(bt_terrain_view_config, morph_range) == 56 && offsetof(bt_terrain_view_config, origin_lod) == 64,
              terrain_view_config);

#define FIELD(type, field) printf("    \"%s\": [%zu, %zu],\n", #field, offsetof(type, field), sizeof(((type*)0)->field))
#define BEGIN(type) printf("  \"%s\": {\n", #type)
#define END(type) printf("    \"__size__\": [0, %zu]\n  },\n", sizeof(type))

typedef uint32_t (*abi_version_fn)(void);
typedef bt_status (*ctx_create_fn)(int32_t, void*, bt_ctx**);
typedef void (*ctx_destroy_fn)(bt_ctx*);
typedef const char* (*last_error_fn)(void);
typedef bt_status (*view_state_fn)(const bt_terrain_model*, const bt_terrain_view_config*, const double*, float, bt_view_state*);
typedef void (*view_config_default_fn)(bt_terrain_view_config*);

int main(int argc, char** argv) {
    printf("{\n");
    BEGIN(bt_tile_coordinate); FIELD(bt_tile_coordinate, side); FIELD(bt_tile_coordinate, lod); FIELD(bt_tile_coordinate, x); FIELD(bt_tile_coordinate, y); END(bt_tile_coordinate);
    BEGIN(bt_atlas_tile); FIELD(bt_atlas_tile, coordinate); FIELD(bt_atlas_tile, atlas_index); FIELD(bt_atlas_tile, _padding); END(bt_atlas_tile);
    BEGIN(bt_attachment_config); FIELD(bt_attachment_config, name); FIELD(bt_attachment_config, texture_size); FIELD(bt_attachment_config, border_size); FIELD(bt_attachment_config, mip_level_count); FIELD(bt_attachment_config, format); END(bt_attachment_config);
    BEGIN(bt_terrain_config); FIELD(bt_terrain_config, lod_count); FIELD(bt_terrain_config, atlas_size); FIELD(bt_terrain_config, spherical); FIELD(bt_terrain_config, attachment_count); FIELD(bt_terrain_config, attachments); FIELD(bt_terrain_config, path); END(bt_terrain_config);
    BEGIN(bt_raster); FIELD(bt_raster, data); FIELD(bt_raster, width); FIELD(bt_raster, height); FIELD(bt_raster, row_pitch); FIELD(bt_raster, format); FIELD(bt_raster, on_device); END(bt_raster);
    BEGIN(bt_preprocess_dataset); FIELD(bt_preprocess_dataset, attachment_index); FIELD(bt_preprocess_dataset, side); FIELD(bt_preprocess_dataset, top_left); FIELD(bt_preprocess_dataset, bottom_right); FIELD(bt_preprocess_dataset, lod_begin); FIELD(bt_preprocess_dataset, lod_end); END(bt_preprocess_dataset);
    BEGIN(bt_spherical_dataset); FIELD(bt_spherical_dataset, attachment_index); FIELD(bt_spherical_dataset, lod_begin); FIELD(bt_spherical_dataset, lod_end); END(bt_spherical_dataset);
    BEGIN(bt_tile_tree_entry); FIELD(bt_tile_tree_entry, atlas_index); FIELD(bt_tile_tree_entry, atlas_lod); END(bt_tile_tree_entry);
    BEGIN(bt_run_stats); FIELD(bt_run_stats, kernel_launches); FIELD(bt_run_stats, tiles); FIELD(bt_run_stats, algorithmic_bytes); FIELD(bt_run_stats, fused_jobs); FIELD(bt_run_stats, generic_jobs); FIELD(bt_run_stats, prev_zero_launches); FIELD(bt_run_stats, reserved); END(bt_run_stats);
    BEGIN(bt_stream_stats); FIELD(bt_stream_stats, streamed); FIELD(bt_stream_stats, bands); FIELD(bt_stream_stats, banded_launches); FIELD(bt_stream_stats, early_tiles); FIELD(bt_stream_stats, uploaded_bytes); FIELD(bt_stream_stats, saved_bytes); END(bt_stream_stats);
    BEGIN(bt_shard_range); FIELD(bt_shard_range, attachment_index); FIELD(bt_shard_range, side); FIELD(bt_shard_range, lod); FIELD(bt_shard_range, first_layer); FIELD(bt_shard_range, layers_per_rank); END(bt_shard_range);
    BEGIN(bt_launch_profile); FIELD(bt_launch_profile, kind); FIELD(bt_launch_profile, tasks); FIELD(bt_launch_profile, algorithmic_bytes); FIELD(bt_launch_profile, avg_ms); FIELD(bt_launch_profile, samples); END(bt_launch_profile);
    BEGIN(bt_side_parameter); FIELD(bt_side_parameter, view_xy); FIELD(bt_side_parameter, view_uv); END(bt_side_parameter);
    BEGIN(bt_view_state); FIELD(bt_view_state, spherical); FIELD(bt_view_state, geometry_tile_count); FIELD(bt_view_state, refinement_count); FIELD(bt_view_state, vertices_per_tile); FIELD(bt_view_state, subdivision_distance); FIELD(bt_view_state, origin_lod); FIELD(bt_view_state, approximate_height); FIELD(bt_view_state, sides); FIELD(bt_view_state, world_position); FIELD(bt_view_state, world_from_local); FIELD(bt_view_state, local_from_world_transpose); END(bt_view_state);
    BEGIN(bt_indirect); FIELD(bt_indirect, vertex_count); FIELD(bt_indirect, instance_count); FIELD(bt_indirect, base_vertex); FIELD(bt_indirect, base_instance); END(bt_indirect);
    BEGIN(bt_terrain_model); FIELD(bt_terrain_model, kind); FIELD(bt_terrain_model, _padding); FIELD(bt_terrain_model, position); FIELD(bt_terrain_model, a); FIELD(bt_terrain_model, b); FIELD(bt_terrain_model, min_height); FIELD(bt_terrain_model, max_height); END(bt_terrain_model);
    BEGIN(bt_terrain_view_config); FIELD(bt_terrain_view_config, tree_size); FIELD(bt_terrain_view_config, geometry_tile_count); FIELD(bt_terrain_view_config, refinement_count); FIELD(bt_terrain_view_config, grid_size); FIELD(bt_terrain_view_config, subdivision_tolerance); FIELD(bt_terrain_view_config, precision_threshold_distance); FIELD(bt_terrain_view_config, load_distance); FIELD(bt_terrain_view_config, morph_distance); FIELD(bt_terrain_view_config, blend_distance); FIELD(bt_terrain_view_config, morph_range); FIELD(bt_terrain_view_config, blend_range); FIELD(bt_terrain_view_config, origin_lod); FIELD(bt_terrain_view_config, _padding); END(bt_terrain_view_config);

    int rc = 0;
    if (argc > 1) {
        void* lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
        if (!lib) {
            fprintf(stderr, "dlopen: %s\n", dlerror());
            return 2;
        }
        abi_version_fn version = (abi_version_fn)dlsym(lib, "bt_abi_version");
        ctx_create_fn create = (ctx_create_fn)dlsym(lib, "bt_ctx_create");
        ctx_destroy_fn destroy = (ctx_destroy_fn)dlsym(lib, "bt_ctx_destroy");
        last_error_fn last_error = (last_error_fn)dlsym(lib, "bt_last_error");
        view_state_fn view_state = (view_state_fn)dlsym(lib, "bt_view_state_from_config");
        view_config_default_fn view_default = (view_config_default_fn)dlsym(lib, "bt_terrain_view_config_default");
        if (!version || !create || !destroy || !last_error || !view_state || !view_default) return 3;
        bt_ctx* ctx = NULL;
        const bt_status status = create(0, NULL, &ctx);
        /* a pure host call through struct arguments: examples/minimal.rs' planar terrain, default view */
        bt_terrain_model model;
        bt_terrain_view_config config;
        bt_view_state view;
        const double position[3] = {100.0, 300.0, -200.0};
        memset(&model, 0, sizeof model);
        model.kind = BT_MODEL_PLANAR;
        model.a = 1000.0;
        model.max_height = 250.0f;
        view_default(&config);
        const bt_status vs = view_state(&model, &config, position, 125.0f, &view);
        printf("  \"abi_version\": %u,\n  \"ctx_create\": %d,\n  \"ctx_create_error\": \"%s\",\n  \"view_state\": [%d, %d, %d, %u, %u],\n", version(), status,
               status ? "set" : "", vs, view.sides[0].view_xy[0], view.sides[0].view_xy[1], view.vertices_per_tile, config.tree_size);
        if (status == BT_OK) destroy(ctx);
        else if (!last_error()[0]) rc = 4; /* an error status must come with a message */
        if (version() != BT_ABI_VERSION) rc = 5;
        dlclose(lib);
    }
    printf("  \"done\": true\n}\n");
    return rc;
}
