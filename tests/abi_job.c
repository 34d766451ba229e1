/* A C99 host that RUNS the path through the C ABI — the compiled stand-in for the Rust `extern "C"` block of INTEGRATION.md §1
 * (no rustc in this image).  It does what TerrainPreprocessNode::run + the Save tasks (src/preprocess/mod.rs:143-218,
 * preprocess/preprocessor.rs:346-399) and TilingPrepassNode::run (src/render/tiling_prepass.rs:204-272) make the reference do:
 *
 *   ctx + atlas (TerrainConfig: lod_count 3, one R16 attachment "height", T = 64, b = 2)
 *   -> bt_preprocessor_clear_attachment -> bt_preprocessor_preprocess_tile on a 300 x 300 HOST raster (a formula the test
 *      repeats in numpy; a no-data patch inside)  -> bt_preprocessor_run -> bt_preprocessor_save  (.bin tiles + config.tc)
 *   -> bt_view_state_from_config -> bt_tiling_prepass_run -> bt_tiling_prepass_read            (final tile list + indirect args)
 *
 * and writes the tile list to "<out>/final_tiles.bin" (u32 count, bt_indirect, count x bt_tile_coordinate).
 * tests/test_gpu_abi_job.py compiles it with gcc, runs it on the GPU and compares every file with the oracle's.
 *   gcc -std=c99 -Wall -Wextra -Werror -Iinclude tests/abi_job.c -ldl -o abi_job && ./abi_job lib.so outdir */
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "bevy_terrain_amd.h"

#define W 300u
#define H 300u

/* one pointer per entry point used, resolved by name: exactly what a foreign-function binding does */
#define ENTRY_POINTS(X)                                                                                                                          \
    X(bt_last_error) X(bt_ctx_create) X(bt_ctx_destroy) X(bt_ctx_synchronize) X(bt_atlas_create) X(bt_atlas_destroy) X(bt_atlas_tiles)            \
    X(bt_preprocessor_create) X(bt_preprocessor_destroy) X(bt_preprocessor_clear_attachment) X(bt_preprocessor_preprocess_tile)                  \
    X(bt_preprocessor_run) X(bt_preprocessor_save) X(bt_preprocessor_last_run_stats) X(bt_terrain_view_config_default)                           \
    X(bt_view_state_from_config) X(bt_tiling_prepass_create) X(bt_tiling_prepass_destroy) X(bt_tiling_prepass_run) X(bt_tiling_prepass_read)
#define DECLARE(name) static __typeof__(&name) p_##name;
ENTRY_POINTS(DECLARE)

static int check(bt_status s, const char* what) {
    if (s == BT_OK) return 0;
    fprintf(stderr, "%s: status %d: %s\n", what, s, p_bt_last_error());
    return 1;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: abi_job libbevy_terrain_amd.so out_dir\n");
        return 2;
    }
    void* lib = dlopen(argv[1], RTLD_NOW | RTLD_GLOBAL);
    if (!lib) {
        fprintf(stderr, "dlopen: %s\n", dlerror());
        return 2;
    }
#define RESOLVE(name)                                                \
    *(void**)(&p_##name) = dlsym(lib, #name);                        \
    if (!p_##name) {                                                 \
        fprintf(stderr, "missing symbol %s\n", #name);               \
        return 3;                                                    \
    }
    ENTRY_POINTS(RESOLVE)

    /* the source raster: values 1 .. 60000, a 20 x 30 no-data patch */
    uint16_t* raster = (uint16_t*)malloc(sizeof(uint16_t) * W * H);
    for (uint32_t y = 0; y < H; y++)
        for (uint32_t x = 0; x < W; x++) raster[y * W + x] = (uint16_t)(1u + (x * 131u + y * 71u + (x * y) % 97u) % 60000u);
    for (uint32_t y = 40; y < 60; y++)
        for (uint32_t x = 100; x < 130; x++) raster[y * W + x] = 0;

    bt_ctx* ctx = NULL;
    bt_atlas* atlas = NULL;
    bt_preprocessor* pre = NULL;
    bt_tiling_prepass* prepass = NULL;
    int rc = 1;
    if (check(p_bt_ctx_create(0, NULL, &ctx), "bt_ctx_create")) return 4;

    bt_terrain_config config;
    memset(&config, 0, sizeof config);
    config.lod_count = 3;
    config.atlas_size = 64;
    config.attachment_count = 1;
    strcpy(config.attachments[0].name, "height");
    config.attachments[0].texture_size = 64;
    config.attachments[0].border_size = 2;
    config.attachments[0].mip_level_count = 1;
    config.attachments[0].format = BT_FORMAT_R16;
    strcpy(config.path, "terrains/abi_job");
    if (check(p_bt_atlas_create(ctx, &config, &atlas), "bt_atlas_create")) goto done;
    if (check(p_bt_preprocessor_create(ctx, &pre), "bt_preprocessor_create")) goto done;
    {   /* Preprocessor::clear_attachment + preprocess_tile (preprocessor.rs:290-312) */
        char dir[1024];
        snprintf(dir, sizeof dir, "%s/terrains/abi_job/data/height", argv[2]);
        if (check(p_bt_preprocessor_clear_attachment(pre, atlas, 0, dir), "bt_preprocessor_clear_attachment")) goto done;
        bt_preprocess_dataset dataset;
        memset(&dataset, 0, sizeof dataset);
        dataset.attachment_index = 0;
        dataset.side = 0;
        dataset.top_left[0] = dataset.top_left[1] = 0.0f;
        dataset.bottom_right[0] = dataset.bottom_right[1] = 1.0f;
        dataset.lod_begin = 0;
        dataset.lod_end = 3;
        bt_raster source;
        memset(&source, 0, sizeof source);
        source.data = raster;
        source.width = W;
        source.height = H;
        source.row_pitch = W * sizeof(uint16_t);
        source.format = BT_FORMAT_R16;
        source.on_device = 0;
        if (check(p_bt_preprocessor_preprocess_tile(pre, atlas, &dataset, &source), "bt_preprocessor_preprocess_tile")) goto done;
    }
    if (check(p_bt_preprocessor_run(pre, atlas, BT_RUN_AUTO | BT_RUN_KEEP_QUEUE), "bt_preprocessor_run")) goto done;
    bt_run_stats stats;
    if (check(p_bt_preprocessor_last_run_stats(pre, &stats), "bt_preprocessor_last_run_stats")) goto done;
    if (check(p_bt_preprocessor_save(pre, atlas, argv[2]), "bt_preprocessor_save")) goto done;
    printf("{\"tiles\": %u, \"kernel_launches\": %u, \"fused_jobs\": %u, \"atlas_tiles\": %u, ", stats.tiles, stats.kernel_launches, stats.fused_jobs,
           p_bt_atlas_tiles(atlas, NULL, NULL, 0));

    {   /* the tiling prepass of one view (examples/minimal.rs' planar terrain seen from above) */
        bt_terrain_model model;
        bt_terrain_view_config view_config;
        bt_view_state view;
        const double position[3] = {120.0, 260.0, -75.0};
        memset(&model, 0, sizeof model);
        model.kind = BT_MODEL_PLANAR;
        model.a = 1000.0;
        model.max_height = 250.0f;
        p_bt_terrain_view_config_default(&view_config);
        view_config.geometry_tile_count = 100000;
        if (check(p_bt_view_state_from_config(&model, &view_config, position, 0.0f, &view), "bt_view_state_from_config")) goto done;
        if (check(p_bt_tiling_prepass_create(ctx, view_config.geometry_tile_count, &prepass), "bt_tiling_prepass_create")) goto done;
        if (check(p_bt_tiling_prepass_run(prepass, &view), "bt_tiling_prepass_run")) goto done;
        bt_tile_coordinate* tiles = (bt_tile_coordinate*)malloc(sizeof(bt_tile_coordinate) * view_config.geometry_tile_count);
        uint32_t count = 0;
        bt_indirect indirect;
        if (check(p_bt_tiling_prepass_read(prepass, tiles, view_config.geometry_tile_count, &count, &indirect), "bt_tiling_prepass_read")) {
            free(tiles);
            goto done;
        }
        char path[1024];
        snprintf(path, sizeof path, "%s/final_tiles.bin", argv[2]);
        FILE* f = fopen(path, "wb");
        if (!f) {
            free(tiles);
            goto done;
        }
        fwrite(&count, sizeof count, 1, f);
        fwrite(&indirect, sizeof indirect, 1, f);
        fwrite(tiles, sizeof(bt_tile_coordinate), count, f);
        fclose(f);
        free(tiles);
        printf("\"final_tiles\": %u, \"vertex_count\": %u}\n", count, indirect.vertex_count);
    }
    rc = 0;
done:
    if (prepass) p_bt_tiling_prepass_destroy(prepass);
    if (pre) p_bt_preprocessor_destroy(pre);
    if (atlas) p_bt_atlas_destroy(atlas);
    p_bt_ctx_destroy(ctx);
    free(raster);
    dlclose(lib);
    return rc;
}
