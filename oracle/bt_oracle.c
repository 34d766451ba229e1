/*
 * bt_oracle.c — CPU ORACLE.  TEST INFRASTRUCTURE ONLY (see bt_oracle.h).
 * PARITY UNPINNED: the reference has no golden vectors; this is a restatement.
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math -fopenmp -shared -fPIC
 * All float arithmetic is IEEE binary32, one rounding per operation, in the
 * order written (no FMA contraction).
 *
 * Citations are relative to /root/reference/ (kurtkuehnert/bevy_terrain).
 */
#include "bt_oracle.h"

#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ======================================================================== */
/* math/coordinate.rs                                                       */
/* ======================================================================== */

static const orc_coord COORD_INVALID = {ORC_INVALID, ORC_INVALID, ORC_INVALID, ORC_INVALID};

int orc_coord_is_invalid(orc_coord c) {
    return c.side == ORC_INVALID && c.lod == ORC_INVALID && c.x == ORC_INVALID && c.y == ORC_INVALID;
}

static int coord_eq(orc_coord a, orc_coord b) {
    return a.side == b.side && a.lod == b.lod && a.x == b.x && a.y == b.y;
}

/* coordinate.rs:9-16 */
static const uint32_t NEIGHBOURING_SIDES[6][5] = {
    {0, 4, 2, 1, 5}, {1, 0, 2, 3, 5}, {2, 0, 4, 3, 1}, {3, 2, 4, 5, 1}, {4, 2, 0, 5, 3}, {5, 4, 0, 1, 3},
};

/* coordinate.rs:18-53  SideInfo */
enum { SI_FIXED0, SI_FIXED1, SI_POS_S, SI_POS_T };
static const int SI_EVEN_LIST[6][2] = {
    {SI_POS_S, SI_POS_T}, {SI_FIXED0, SI_POS_T}, {SI_FIXED0, SI_POS_S},
    {SI_POS_T, SI_POS_S}, {SI_POS_T, SI_FIXED0}, {SI_POS_S, SI_FIXED0},
};
static const int SI_ODD_LIST[6][2] = {
    {SI_POS_S, SI_POS_T}, {SI_POS_S, SI_FIXED1}, {SI_POS_T, SI_FIXED1},
    {SI_POS_T, SI_POS_S}, {SI_FIXED1, SI_POS_S}, {SI_FIXED1, SI_POS_T},
};

static const int* side_info_project(uint32_t side, uint32_t other_side) {
    uint32_t index = (6 + other_side - side) % 6; /* coordinate.rs:45 */
    return (side % 2 == 0) ? SI_EVEN_LIST[index] : SI_ODD_LIST[index];
}

/* coordinate.rs:187-194 */
orc_coord orc_parent(orc_coord c) {
    orc_coord p = {c.side, c.lod - 1u, c.x >> 1, c.y >> 1};
    return p;
}

/* coordinate.rs:196-206 */
void orc_children(orc_coord c, orc_coord out[4]) {
    for (uint32_t index = 0; index < 4; index++) {
        out[index].side = c.side;
        out[index].lod = c.lod + 1;
        out[index].x = (c.x << 1) + index % 2;
        out[index].y = (c.y << 1) + index / 2;
    }
}

/* coordinate.rs:226-279 */
static orc_coord neighbour_coordinate(orc_coord self, int nx, int ny, int spherical) {
    int tile_count = (int)(1u << self.lod);

    if (spherical) {
        int edge_index;
        if ((nx < 0 && ny < 0) || (nx < 0 && ny >= tile_count) || (nx >= tile_count && ny < 0) ||
            (nx >= tile_count && ny >= tile_count)) {
            return COORD_INVALID;
        } else if (nx < 0) {
            edge_index = 1;
        } else if (ny < 0) {
            edge_index = 2;
        } else if (nx >= tile_count) {
            edge_index = 3;
        } else if (ny >= tile_count) {
            edge_index = 4;
        } else {
            edge_index = 0;
        }

        int cx = nx < 0 ? 0 : (nx > tile_count - 1 ? tile_count - 1 : nx);
        int cy = ny < 0 ? 0 : (ny > tile_count - 1 ? tile_count - 1 : ny);

        uint32_t neighbour_side = NEIGHBOURING_SIDES[self.side][edge_index];
        const int* info = side_info_project(self.side, neighbour_side);

        uint32_t xy[2];
        for (int k = 0; k < 2; k++) {
            switch (info[k]) {
                case SI_FIXED0: xy[k] = 0; break;
                case SI_FIXED1: xy[k] = (uint32_t)tile_count - 1; break;
                case SI_POS_S: xy[k] = (uint32_t)cx; break;
                default: xy[k] = (uint32_t)cy; break;
            }
        }
        orc_coord r = {neighbour_side, self.lod, xy[0], xy[1]};
        return r;
    } else {
        if (nx < 0 || ny < 0 || nx >= tile_count || ny >= tile_count) return COORD_INVALID;
        orc_coord r = {self.side, self.lod, (uint32_t)nx, (uint32_t)ny};
        return r;
    }
}

/* coordinate.rs:208-224 */
void orc_neighbours(orc_coord c, int spherical, orc_coord out[8]) {
    static const int OFFSETS[8][2] = {{0, -1}, {1, 0}, {0, 1}, {-1, 0}, {-1, -1}, {1, -1}, {1, 1}, {-1, 1}};
    for (int i = 0; i < 8; i++)
        out[i] = neighbour_coordinate(c, (int)c.x + OFFSETS[i][0], (int)c.y + OFFSETS[i][1], spherical);
}

int orc_coord_name(orc_coord c, char* buf, size_t n) {
    return snprintf(buf, n, "%u_%u_%u_%u", c.side, c.lod, c.x, c.y);
}

/* ======================================================================== */
/* atlas state: terrain_data/tile_atlas.rs:279-416                          */
/* ======================================================================== */

typedef struct {
    orc_coord coord;
    uint32_t atlas_index;
    int used;     /* slot occupied: entry of tile_states */
    int existing; /* member of existing_tiles */
} tile_entry;

typedef struct {
    orc_attachment_config cfg;
    uint32_t center_size;
    uint32_t pixel_size;
    uint8_t* data; /* atlas_size tiles, T*T*pixel_size bytes each, zero on creation (wgpu zero-inits) */
} attachment;

enum { TASK_SPLIT, TASK_STITCH, TASK_DOWNSAMPLE, TASK_SAVE, TASK_BARRIER };

typedef struct {
    int type;
    orc_coord coord;
    uint32_t atlas_index;
    uint32_t attachment_index;
    orc_atlas_tile rel[8]; /* children (4) or neighbours (8) */
    float top_left[2], bottom_right[2];
    const void* src;
    uint32_t src_w, src_h;
} task;

struct orc_atlas {
    uint32_t lod_count, atlas_size, n_att;
    int spherical;
    attachment att[8];
    tile_entry* table; /* open addressing */
    uint32_t table_cap;
    uint32_t next_unused; /* unused_tiles is 0..atlas_size FIFO: tile_atlas.rs:307-309 */
    task* tasks;
    size_t n_tasks, cap_tasks;
    orc_task_backend backend; /* NULL: the kernels below */
    void* backend_user;
};

static uint64_t coord_hash(orc_coord c) {
    uint64_t h = 1469598103934665603ull;
    uint32_t v[4] = {c.side, c.lod, c.x, c.y};
    for (int i = 0; i < 4; i++) {
        h ^= v[i];
        h *= 1099511628211ull;
        h ^= h >> 29;
    }
    return h;
}

static tile_entry* table_find(const orc_atlas* a, orc_coord c, int insert) {
    uint32_t mask = a->table_cap - 1;
    uint32_t i = (uint32_t)coord_hash(c) & mask;
    for (;;) {
        tile_entry* e = &a->table[i];
        if (!e->used) {
            if (!insert) return NULL;
            e->used = 1;
            e->coord = c;
            e->atlas_index = ORC_INVALID;
            e->existing = 0;
            return e;
        }
        if (coord_eq(e->coord, c)) return e;
        i = (i + 1) & mask;
    }
}

orc_atlas* orc_atlas_new(uint32_t lod_count, uint32_t atlas_size, int spherical, uint32_t n_attachments,
                         const orc_attachment_config* att) {
    if (n_attachments > 8) return NULL;
    orc_atlas* a = (orc_atlas*)calloc(1, sizeof(orc_atlas));
    a->lod_count = lod_count;
    a->atlas_size = atlas_size;
    a->spherical = spherical;
    a->n_att = n_attachments;
    for (uint32_t i = 0; i < n_attachments; i++) {
        a->att[i].cfg = att[i];
        a->att[i].center_size = att[i].texture_size - 2 * att[i].border_size; /* tile_atlas.rs:176 */
        a->att[i].pixel_size = att[i].format == ORC_FORMAT_R16 ? 2 : 4;       /* mod.rs:77-84 */
        size_t bytes = (size_t)atlas_size * att[i].texture_size * att[i].texture_size * a->att[i].pixel_size;
        a->att[i].data = (uint8_t*)calloc(bytes ? bytes : 1, 1);
        if (!a->att[i].data) return NULL;
    }
    uint32_t cap = 64;
    while (cap < atlas_size * 4u + 64u) cap <<= 1;
    a->table_cap = cap;
    a->table = (tile_entry*)calloc(cap, sizeof(tile_entry));
    return a;
}

void orc_atlas_free(orc_atlas* a) {
    if (!a) return;
    for (uint32_t i = 0; i < a->n_att; i++) free(a->att[i].data);
    free(a->table);
    free(a->tasks);
    free(a);
}

/* tile_atlas.rs:369-381 */
static orc_atlas_tile atlas_get_tile(const orc_atlas* a, orc_coord c) {
    orc_atlas_tile t;
    memset(&t, 0, sizeof t);
    t.coordinate = c;
    t.atlas_index = ORC_INVALID;
    if (orc_coord_is_invalid(c)) return t;
    tile_entry* e = table_find(a, c, 0);
    if (e && e->existing) t.atlas_index = e->atlas_index;
    return t;
}

/* tile_atlas.rs:383-416 ; returns -1 on "Atlas out of indices" */
static int atlas_get_or_allocate(orc_atlas* a, orc_coord c, orc_atlas_tile* out) {
    memset(out, 0, sizeof *out);
    out->coordinate = c;
    out->atlas_index = ORC_INVALID;
    if (orc_coord_is_invalid(c)) return 0;
    tile_entry* e = table_find(a, c, 1);
    e->existing = 1;
    if (e->atlas_index == ORC_INVALID) {
        if (a->next_unused >= a->atlas_size) return -1;
        e->atlas_index = a->next_unused++;
    }
    out->atlas_index = e->atlas_index;
    return 0;
}

uint32_t orc_get_tile(const orc_atlas* a, orc_coord c) { return atlas_get_tile(a, c).atlas_index; }

/* preprocessor.rs:290-296 (directory reset is done by the caller of orc_save_*) */
void orc_clear_attachment(orc_atlas* a, uint32_t attachment_index) {
    (void)attachment_index;
    for (uint32_t i = 0; i < a->table_cap; i++) a->table[i].existing = 0;
}

uint32_t orc_tile_count(const orc_atlas* a) {
    uint32_t n = 0;
    for (uint32_t i = 0; i < a->table_cap; i++) n += a->table[i].used && a->table[i].existing;
    return n;
}

uint32_t orc_tiles(const orc_atlas* a, orc_coord* coords, uint32_t* atlas_indices, uint32_t cap) {
    /* atlas-index order = allocation order */
    uint32_t n = 0;
    for (uint32_t idx = 0; idx < a->next_unused; idx++) {
        for (uint32_t i = 0; i < a->table_cap; i++) {
            const tile_entry* e = &a->table[i];
            if (e->used && e->existing && e->atlas_index == idx) {
                if (n < cap) {
                    if (coords) coords[n] = e->coord;
                    if (atlas_indices) atlas_indices[n] = idx;
                }
                n++;
            }
        }
    }
    return n;
}

const void* orc_tile_data(const orc_atlas* a, uint32_t ai, uint32_t atlas_index) {
    const attachment* at = &a->att[ai];
    return at->data + (size_t)atlas_index * at->cfg.texture_size * at->cfg.texture_size * at->pixel_size;
}

size_t orc_tile_bytes(const orc_atlas* a, uint32_t ai) {
    const attachment* at = &a->att[ai];
    return (size_t)at->cfg.texture_size * at->cfg.texture_size * at->pixel_size;
}

/* ======================================================================== */
/* queue construction: preprocess/preprocessor.rs                            */
/* ======================================================================== */

static task* push_task(orc_atlas* a) {
    if (a->n_tasks == a->cap_tasks) {
        a->cap_tasks = a->cap_tasks ? a->cap_tasks * 2 : 256;
        a->tasks = (task*)realloc(a->tasks, a->cap_tasks * sizeof(task));
    }
    task* t = &a->tasks[a->n_tasks++];
    memset(t, 0, sizeof *t);
    return t;
}

static void push_barrier(orc_atlas* a) { push_task(a)->type = TASK_BARRIER; } /* :128-133 */

/* preprocessor.rs:58-66 : x outer, y inner; as_uvec2 truncates, ceil() before the upper cast */
typedef struct {
    uint32_t lx, ly, ux, uy;
} tile_range;

/* `as_uvec2` (glam: `as u32` per component) saturates: negative and NaN -> 0, too large -> u32::MAX; a plain C cast of a negative
 * float is undefined.  DEVIATION shared with the product (deliberate deviations): the range is then clamped to the
 * face — a bottom_right > 1 would make the reference queue tiles with x, y >= 2^lod, which do not exist (found by the second
 * model, tests/_second_models.py, round 4: until then this function cast without saturating and did not clamp). */
static uint32_t as_u32_saturating(float v, uint32_t face) {
    if (!(v > 0.0f)) return 0u;
    if (v >= (float)face) return face;
    return (uint32_t)v;
}
static tile_range overlapping_tiles(const orc_dataset* d, uint32_t lod) {
    float tile_count = (float)(1u << lod);
    tile_range r;
    r.lx = as_u32_saturating(d->top_left[0] * tile_count, 1u << lod);
    r.ly = as_u32_saturating(d->top_left[1] * tile_count, 1u << lod);
    r.ux = as_u32_saturating(ceilf(d->bottom_right[0] * tile_count), 1u << lod);
    r.uy = as_u32_saturating(ceilf(d->bottom_right[1] * tile_count), 1u << lod);
    return r;
}

/* preprocessor.rs:234-269 */
static int split_and_downsample(orc_atlas* a, const orc_dataset* d, const void* src, uint32_t w, uint32_t h) {
    if (d->lod_end <= d->lod_begin) return -2;
    uint32_t lod = d->lod_end - 1; /* lods = lod_range.rev(); first = finest */
    tile_range r = overlapping_tiles(d, lod);
    for (uint32_t x = r.lx; x < r.ux; x++)
        for (uint32_t y = r.ly; y < r.uy; y++) {
            orc_coord c = {d->side, lod, x, y};
            orc_atlas_tile t;
            if (atlas_get_or_allocate(a, c, &t)) return -1;
            task* k = push_task(a); /* PreprocessTask::split :153-172 */
            k->type = TASK_SPLIT;
            k->coord = c;
            k->atlas_index = t.atlas_index;
            k->attachment_index = d->attachment_index;
            k->top_left[0] = d->top_left[0];
            k->top_left[1] = d->top_left[1];
            k->bottom_right[0] = d->bottom_right[0];
            k->bottom_right[1] = d->bottom_right[1];
            k->src = src;
            k->src_w = w;
            k->src_h = h;
        }
    while (lod > d->lod_begin) {
        lod--;
        push_barrier(a);
        r = overlapping_tiles(d, lod);
        for (uint32_t x = r.lx; x < r.ux; x++)
            for (uint32_t y = r.ly; y < r.uy; y++) {
                orc_coord c = {d->side, lod, x, y};
                orc_atlas_tile t;
                if (atlas_get_or_allocate(a, c, &t)) return -1;
                task* k = push_task(a); /* PreprocessTask::downsample :191-210 */
                k->type = TASK_DOWNSAMPLE;
                k->coord = c;
                k->atlas_index = t.atlas_index;
                k->attachment_index = d->attachment_index;
                orc_coord ch[4];
                orc_children(c, ch);
                for (int i = 0; i < 4; i++) k->rel[i] = atlas_get_tile(a, ch[i]);
            }
    }
    return 0;
}

/* preprocessor.rs:271-288 */
static int stitch_and_save_layer(orc_atlas* a, const orc_dataset* d, uint32_t lod) {
    tile_range r = overlapping_tiles(d, lod);
    for (uint32_t x = r.lx; x < r.ux; x++)
        for (uint32_t y = r.ly; y < r.uy; y++) {
            orc_coord c = {d->side, lod, x, y};
            orc_atlas_tile t;
            if (atlas_get_or_allocate(a, c, &t)) return -1;
            task* k = push_task(a); /* PreprocessTask::stitch :170-189 */
            k->type = TASK_STITCH;
            k->coord = c;
            k->atlas_index = t.atlas_index;
            k->attachment_index = d->attachment_index;
            orc_coord nb[8];
            orc_neighbours(c, a->spherical, nb);
            for (int i = 0; i < 8; i++) k->rel[i] = atlas_get_tile(a, nb[i]);
        }
    push_barrier(a);
    for (uint32_t x = r.lx; x < r.ux; x++)
        for (uint32_t y = r.ly; y < r.uy; y++) {
            orc_coord c = {d->side, lod, x, y};
            orc_atlas_tile t;
            if (atlas_get_or_allocate(a, c, &t)) return -1;
            task* k = push_task(a); /* PreprocessTask::save :135-151 */
            k->type = TASK_SAVE;
            k->coord = c;
            k->atlas_index = t.atlas_index;
            k->attachment_index = d->attachment_index;
        }
    return 0;
}

/* preprocessor.rs:298-312 */
int orc_preprocess_tile(orc_atlas* a, const orc_dataset* d, const void* src, uint32_t w, uint32_t h) {
    int rc = split_and_downsample(a, d, src, w, h);
    if (rc) return rc;
    push_barrier(a);
    for (uint32_t lod = d->lod_begin; lod < d->lod_end; lod++) {
        rc = stitch_and_save_layer(a, d, lod);
        if (rc) return rc;
    }
    return 0;
}

/* preprocessor.rs:314-343 */
int orc_preprocess_spherical(orc_atlas* a, uint32_t attachment_index, uint32_t lod_begin, uint32_t lod_end,
                             const void* const src[6], uint32_t w, uint32_t h) {
    orc_dataset side_datasets[6];
    for (uint32_t side = 0; side < 6; side++) {
        orc_dataset* d = &side_datasets[side];
        d->attachment_index = attachment_index;
        d->side = side;
        d->top_left[0] = d->top_left[1] = 0.0f; /* ..default() :44-55 */
        d->bottom_right[0] = d->bottom_right[1] = 1.0f;
        d->lod_begin = lod_begin;
        d->lod_end = lod_end;
    }
    for (uint32_t side = 0; side < 6; side++) {
        int rc = split_and_downsample(a, &side_datasets[side], src[side], w, h);
        if (rc) return rc;
    }
    push_barrier(a);
    for (uint32_t lod = lod_begin; lod < lod_end; lod++)
        for (uint32_t side = 0; side < 6; side++) {
            int rc = stitch_and_save_layer(a, &side_datasets[side], lod);
            if (rc) return rc;
        }
    return 0;
}

uint32_t orc_task_count(const orc_atlas* a, uint32_t counts[5]) {
    if (counts) {
        memset(counts, 0, 5 * sizeof(uint32_t));
        for (size_t i = 0; i < a->n_tasks; i++) counts[a->tasks[i].type]++;
    }
    return (uint32_t)a->n_tasks;
}

/* ======================================================================== */
/* kernels: shaders/preprocess/*.wgsl                                        */
/* ======================================================================== */

typedef struct {
    float v[4];
} vec4;

static const vec4 VEC4_ZERO = {{0.0f, 0.0f, 0.0f, 0.0f}};

/* texel -> float as textureLoad / the sampler sees it.  R16Unorm: (r/65535, 0, 0, 1);
 * Rgba8Unorm: channel/255 (WebGPU unorm conversion, defined here as one correctly rounded
 * f32 division). */
static vec4 load_texel(uint32_t format, const void* image, size_t texel_index) {
    vec4 r;
    if (format == ORC_FORMAT_R16) {
        uint16_t t = ((const uint16_t*)image)[texel_index];
        r.v[0] = (float)t / 65535.0f;
        r.v[1] = 0.0f;
        r.v[2] = 0.0f;
        r.v[3] = 1.0f;
    } else {
        const uint8_t* p = (const uint8_t*)image + texel_index * 4;
        for (int k = 0; k < 4; k++) r.v[k] = (float)p[k] / 255.0f;
    }
    return r;
}

/* textureLoad(atlas, coords, layer, 0); a layer outside the atlas (INVALID_ATLAS_INDEX)
 * reads as zero (WGSL leaves it implementation-defined; zero = "no data"). */
static vec4 atlas_load(const orc_atlas* a, const attachment* at, uint32_t x, uint32_t y, uint32_t layer) {
    uint32_t T = at->cfg.texture_size;
    if (layer >= a->atlas_size || x >= T || y >= T) return VEC4_ZERO;
    const uint8_t* tile = at->data + (size_t)layer * T * T * at->pixel_size;
    return load_texel(at->cfg.format, tile, (size_t)y * T + x);
}

/* preprocessing.wgsl:44-57 */
static int inside_u(uint32_t cx, uint32_t cy, uint32_t bx, uint32_t by, uint32_t bw, uint32_t bh) {
    return cx >= bx && cx < bx + bw && cy >= by && cy < by + bh;
}
static int is_border(uint32_t T, uint32_t b, uint32_t px, uint32_t py) {
    (void)T;
    uint32_t c = T - 2 * b;
    return !inside_u(px, py, b, b, c, c);
}

/* WGSL mix(e1,e2,e3) = e1*(1-e3) + e2*e3, evaluated in f32 without contraction. */
static float mixf(float a, float b, float t) { return a * (1.0f - t) + b * t; }

static int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* textureSampleLevel with ImageSampler::linear() (preprocessor.rs:409): the reference uses the
 * GPU's fixed-function filter, whose weights are implementation-defined.  Defined here as exact
 * f32 bilinear: texel centres at uv*N-0.5, clamp-to-edge, value = mix(mix(t00,t10,fx),
 * mix(t01,t11,fx), fy).  textureGather(0, ..) = channel 0 of the same four footprint texels. */
/* Sampler model (orc_set_sampler_model).  0 bits = the exact-f32 definition above, which is what the product
 * implements bit for bit.  N > 0 restates what a hardware sampler may legally do at split.wgsl:32: the filter weights
 * carry only N fractional bits (D3D / Vulkan require at least 8: VkPhysicalDeviceLimits::subTexelPrecisionBits >= 4,
 * every desktop GPU reports 8).  The weights are snapped to multiples of 2^-N, to nearest (mode 0) or by truncation
 * (mode 1).  This bounds how far the reference, run on a real wgpu device, can sit from the exact definition. */
static int g_weight_bits = 0, g_weight_mode = 0;
void orc_set_sampler_model(int fractional_bits, int mode) {
    g_weight_bits = fractional_bits;
    g_weight_mode = mode;
}
static float snap_weight(float f) {
    if (g_weight_bits <= 0) return f;
    const float scale = (float)(1u << g_weight_bits);
    return floorf(f * scale + (g_weight_mode == 0 ? 0.5f : 0.0f)) / scale;
}

static vec4 sample_bilinear(uint32_t format, const void* src, uint32_t w, uint32_t h, float u, float v,
                            int* all_nonzero) {
    float qx = u * (float)w - 0.5f;
    float qy = v * (float)h - 0.5f;
    float fx0 = floorf(qx), fy0 = floorf(qy);
    float fx = snap_weight(qx - fx0), fy = snap_weight(qy - fy0);
    int ix = (int)fx0, iy = (int)fy0;
    int x0 = clampi(ix, 0, (int)w - 1), x1 = clampi(ix + 1, 0, (int)w - 1);
    int y0 = clampi(iy, 0, (int)h - 1), y1 = clampi(iy + 1, 0, (int)h - 1);
    vec4 t00 = load_texel(format, src, (size_t)y0 * w + x0);
    vec4 t10 = load_texel(format, src, (size_t)y0 * w + x1);
    vec4 t01 = load_texel(format, src, (size_t)y1 * w + x0);
    vec4 t11 = load_texel(format, src, (size_t)y1 * w + x1);
    vec4 r;
    for (int k = 0; k < 4; k++) {
        float top = mixf(t00.v[k], t10.v[k], fx);
        float bot = mixf(t01.v[k], t11.v[k], fx);
        r.v[k] = mixf(top, bot, fy);
    }
    *all_nonzero = t00.v[0] != 0.0f && t10.v[0] != 0.0f && t01.v[0] != 0.0f && t11.v[0] != 0.0f;
    return r;
}

/* functions.wgsl:158-162 */
static float inside_square(float px, float py, float ox, float oy, float size) {
    float ix = (px >= ox ? 1.0f : 0.0f) * (px <= ox + size ? 1.0f : 0.0f);
    float iy = (py >= oy ? 1.0f : 0.0f) * (py <= oy + size ? 1.0f : 0.0f);
    return ix * iy;
}

/* split.wgsl:18-43 */
static vec4 split_pixel_value(uint32_t format, uint32_t T, uint32_t b, orc_coord tile, const float tl[2],
                              const float br[2], const void* src, uint32_t w, uint32_t h, uint32_t px,
                              uint32_t py, vec4 previous) {
    if (is_border(T, b, px, py)) return VEC4_ZERO;
    uint32_t c = T - 2 * b;

    float tile_offset_x = (float)tile.x, tile_offset_y = (float)tile.y;
    float tile_coords_x = (float)(px - b) / (float)c;
    float tile_coords_y = (float)(py - b) / (float)c;
    float tile_scale = (float)(1u << tile.lod); /* functions.wgsl:156 tile_count */

    float sx = (tile_offset_x + tile_coords_x) / tile_scale;
    float sy = (tile_offset_y + tile_coords_y) / tile_scale;

    /* preprocessing.wgsl:40-42 inverse_mix */
    sx = (sx - tl[0]) / (br[0] - tl[0]);
    sy = (sy - tl[1]) / (br[1] - tl[1]);

    int is_valid;
    vec4 value = sample_bilinear(format, src, w, h, sx, sy, &is_valid);
    int is_inside = inside_square(tile_coords_x, tile_coords_y, 0.0f, 0.0f, 1.0f) == 1.0f;

    if (is_valid && is_inside) return value;
    return previous; /* textureLoad(atlas, coords, atlas_index, 0) */
}

/* downsample.wgsl:12-40 */
static vec4 downsample_pixel_value(const orc_atlas* a, const attachment* at, const task* k, uint32_t px,
                                   uint32_t py) {
    uint32_t T = at->cfg.texture_size, b = at->cfg.border_size;
    if (is_border(T, b, px, py)) return VEC4_ZERO;

    uint32_t tx = px - b, ty = py - b;
    uint32_t child_size = at->center_size / 2u;
    uint32_t ccx = 2u * (tx % child_size) + b;
    uint32_t ccy = 2u * (ty % child_size) + b;
    uint32_t child_index = tx / child_size + 2u * (ty / child_size);
    if (child_index > 3) return VEC4_ZERO; /* odd center_size: out of bounds in the reference */

    uint32_t layer = k->rel[child_index].atlas_index;

    static const uint32_t OFFSETS[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
    vec4 value = VEC4_ZERO;
    float count = 0.0f;
    for (int i = 0; i < 4; i++) {
        vec4 cv = atlas_load(a, at, ccx + OFFSETS[i][0], ccy + OFFSETS[i][1], layer);
        int is_valid = cv.v[0] != 0.0f || cv.v[1] != 0.0f || cv.v[2] != 0.0f;
        if (is_valid) {
            for (int c = 0; c < 4; c++) value.v[c] += cv.v[c];
            count += 1.0f;
        }
    }
    /* downsample.wgsl:39 `value / count`: 0/0 is NaN and pack*unorm(NaN) is implementation-defined;
     * defined here as 0 ("no data"), the rule of terrain_data/mod.rs:190-194. */
    if (count == 0.0f) return VEC4_ZERO;
    for (int c = 0; c < 4; c++) value.v[c] = value.v[c] / count;
    return value;
}

/* stitch.wgsl:12-51 */
static void project_to_side(uint32_t T, uint32_t cx, uint32_t cy, uint32_t original_side,
                            uint32_t projected_side, uint32_t* ox, uint32_t* oy) {
    enum { PS = 0, PT = 1, NS = 2, NT = 3 };
    static const int EVEN_LIST[6][2] = {{PS, PT}, {PS, PT}, {NT, PS}, {NT, NS}, {PT, NS}, {PS, PT}};
    static const int ODD_LIST[6][2] = {{PS, PT}, {PS, PT}, {PT, NS}, {PT, PS}, {NT, PS}, {PS, PT}};
    uint32_t index = (6u + projected_side - original_side) % 6u;
    const int* info = (original_side % 2u == 0u) ? EVEN_LIST[index] : ODD_LIST[index];
    uint32_t o[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
        switch (info[k]) {
            case PS: o[k] = cx; break;
            case PT: o[k] = cy; break;
            case NS: o[k] = T - 1u - cx; break;
            default: o[k] = T - 1u - cy; break;
        }
    }
    *ox = o[0];
    *oy = o[1];
}

/* stitch.wgsl:53-75 */
static uint32_t neighbour_index(uint32_t T, uint32_t b, uint32_t px, uint32_t py) {
    uint32_t c = T - 2 * b, o = b + c;
    uint32_t bounds[8][4] = {{b, 0, c, b}, {o, b, b, c}, {b, o, c, b}, {0, b, b, c},
                             {0, 0, b, b}, {o, 0, b, b}, {o, o, b, b}, {0, o, b, b}};
    for (uint32_t i = 0; i < 8; i++)
        if (inside_u(px, py, bounds[i][0], bounds[i][1], bounds[i][2], bounds[i][3])) return i;
    return 0;
}

/* stitch.wgsl:105-118 with neighbour_data :77-96 and repeat_data :98-103 */
static vec4 stitch_pixel_value(const orc_atlas* a, const attachment* at, const task* k, uint32_t px,
                               uint32_t py) {
    uint32_t T = at->cfg.texture_size, b = at->cfg.border_size;
    int c = (int)at->center_size;
    if (!is_border(T, b, px, py)) return atlas_load(a, at, px, py, k->atlas_index);

    uint32_t ni = neighbour_index(T, b, px, py);
    if (k->rel[ni].atlas_index == ORC_INVALID) {
        uint32_t rx = px < b ? b : (px > b + (uint32_t)c - 1u ? b + (uint32_t)c - 1u : px);
        uint32_t ry = py < b ? b : (py > b + (uint32_t)c - 1u ? b + (uint32_t)c - 1u : py);
        return atlas_load(a, at, rx, ry, k->atlas_index);
    }
    const int offsets[8][2] = {{0, c}, {-c, 0}, {0, -c}, {c, 0}, {c, c}, {-c, c}, {-c, -c}, {c, -c}};
    uint32_t qx = (uint32_t)((int)px + offsets[ni][0]);
    uint32_t qy = (uint32_t)((int)py + offsets[ni][1]);
    uint32_t nx, ny;
    project_to_side(T, qx, qy, k->coord.side, k->rel[ni].coordinate.side, &nx, &ny);
    return atlas_load(a, at, nx, ny, k->rel[ni].atlas_index);
}

/* pack2x16unorm / pack4x8unorm component: floor(0.5 + N * clamp(e, 0, 1))  (WGSL spec) */
static uint32_t unorm(float e, float n) {
    float cl = e < 0.0f ? 0.0f : (e > 1.0f ? 1.0f : e);
    return (uint32_t)floorf(0.5f + n * cl);
}

/* preprocessing.wgsl:59-90 : one pixel into the (tight) tile buffer */
static void store_pixel(uint32_t format, uint8_t* tile, uint32_t T, uint32_t px, uint32_t py, vec4 value) {
    if (format == ORC_FORMAT_R16) {
        ((uint16_t*)tile)[(size_t)py * T + px] = (uint16_t)unorm(value.v[0], 65535.0f);
    } else {
        uint8_t* p = tile + ((size_t)py * T + px) * 4;
        for (int k = 0; k < 4; k++) p[k] = (uint8_t)unorm(value.v[k], 255.0f);
    }
}

static void run_task(orc_atlas* a, const task* k, uint8_t* scratch) {
    attachment* at = &a->att[k->attachment_index];
    uint32_t T = at->cfg.texture_size, b = at->cfg.border_size, fmt = at->cfg.format;
    size_t tile_bytes = (size_t)T * T * at->pixel_size;
    uint8_t* dst = at->data + (size_t)k->atlas_index * tile_bytes;
    if (a->backend) {
        orc_task_desc d;
        memset(&d, 0, sizeof d);
        d.type = k->type == TASK_SPLIT ? ORC_TASK_SPLIT : (k->type == TASK_STITCH ? ORC_TASK_STITCH : ORC_TASK_DOWNSAMPLE);
        d.format = fmt;
        d.lod_count = a->lod_count;
        d.texture_size = T;
        d.border_size = b;
        d.atlas_size = a->atlas_size;
        d.tile.coordinate = k->coord;
        d.tile.atlas_index = k->atlas_index;
        memcpy(d.rel, k->rel, sizeof d.rel);
        memcpy(d.top_left, k->top_left, sizeof d.top_left);
        memcpy(d.bottom_right, k->bottom_right, sizeof d.bottom_right);
        d.src = k->src;
        d.src_w = k->src_w;
        d.src_h = k->src_h;
        d.atlas = at->data;
        a->backend(a->backend_user, &d, scratch);
        memcpy(dst, scratch, tile_bytes);
        return;
    }
    /* the reference copies the atlas layer into a write section, lets the kernel overwrite every
     * entry and copies it back (preprocess/mod.rs:169-210): compute into scratch, then store. */
    for (uint32_t py = 0; py < T; py++)
        for (uint32_t px = 0; px < T; px++) {
            vec4 v;
            if (k->type == TASK_SPLIT) {
                vec4 prev = atlas_load(a, at, px, py, k->atlas_index);
                v = split_pixel_value(fmt, T, b, k->coord, k->top_left, k->bottom_right, k->src, k->src_w,
                                      k->src_h, px, py, prev);
            } else if (k->type == TASK_DOWNSAMPLE) {
                v = downsample_pixel_value(a, at, k, px, py);
            } else {
                v = stitch_pixel_value(a, at, k, px, py);
            }
            store_pixel(fmt, scratch, T, px, py, v);
        }
    memcpy(dst, scratch, tile_bytes);
}

void orc_set_task_backend(orc_atlas* a, orc_task_backend backend, void* user) {
    a->backend = backend;
    a->backend_user = user;
}

int orc_run(orc_atlas* a, int threads) {
    size_t max_tile = 0;
    for (uint32_t i = 0; i < a->n_att; i++) {
        size_t tb = orc_tile_bytes(a, i);
        if (tb > max_tile) max_tile = tb;
    }
    if (threads < 1) threads = 1;
    size_t i = 0;
    while (i < a->n_tasks) {
        /* a phase = the tasks up to the next barrier; within a phase tasks only read tiles that the
         * phase does not write (split: source; downsample: children; stitch: centres) */
        size_t j = i;
        while (j < a->n_tasks && a->tasks[j].type != TASK_BARRIER) j++;
        long n = (long)(j - i);
#ifdef _OPENMP
#pragma omp parallel num_threads(threads) if (threads > 1)
#endif
        {
            uint8_t* scratch = (uint8_t*)malloc(max_tile ? max_tile : 1);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 1)
#endif
            for (long t = 0; t < n; t++) {
                const task* k = &a->tasks[i + (size_t)t];
                if (k->type == TASK_SAVE) continue; /* files are written by orc_save_attachment */
                run_task(a, k, scratch);
            }
            free(scratch);
        }
        i = j + 1;
    }
    a->n_tasks = 0;
    return 0;
}

/* ---- the bench's CPU baseline leg ------------------------------------------------------------------------
 * Same tasks, same per-pixel functions, same results as orc_run — scheduled for many cores: the units of a phase are
 * (task, block of `rows_per_block` tile rows) instead of whole tasks, so the top of the pyramid (phases of 64, 16, 4 and 1
 * tasks) still spreads over the machine, and every pixel is written straight into the atlas (all three kernels only read
 * texels the phase does not change: split its own previous texel, downsample the children, stitch the centres).
 * phase_seconds / phase_tasks (cap entries) receive wall time and task count of every phase. */
#include <time.h>
static double now_seconds(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void run_task_rows(orc_atlas* a, const task* k, uint32_t y0, uint32_t y1) {
    attachment* at = &a->att[k->attachment_index];
    uint32_t T = at->cfg.texture_size, b = at->cfg.border_size, fmt = at->cfg.format;
    uint8_t* dst = at->data + (size_t)k->atlas_index * T * T * at->pixel_size;
    for (uint32_t py = y0; py < y1 && py < T; py++)
        for (uint32_t px = 0; px < T; px++) {
            vec4 v;
            if (k->type == TASK_SPLIT) {
                vec4 prev = atlas_load(a, at, px, py, k->atlas_index);
                v = split_pixel_value(fmt, T, b, k->coord, k->top_left, k->bottom_right, k->src, k->src_w, k->src_h, px, py, prev);
            } else if (k->type == TASK_DOWNSAMPLE) {
                v = downsample_pixel_value(a, at, k, px, py);
            } else {
                v = stitch_pixel_value(a, at, k, px, py);
            }
            store_pixel(fmt, dst, T, px, py, v);
        }
}

int orc_run_blocks(orc_atlas* a, int threads, uint32_t rows_per_block, double* phase_seconds, uint32_t* phase_tasks, uint32_t cap,
                   uint32_t* n_phases) {
    if (a->backend || rows_per_block == 0) return -1;
    if (threads < 1) threads = 1;
    uint32_t T_max = 1;
    for (uint32_t i = 0; i < a->n_att; i++)
        if (a->att[i].cfg.texture_size > T_max) T_max = a->att[i].cfg.texture_size;
    const long blocks = (long)((T_max + rows_per_block - 1) / rows_per_block);
    uint32_t phase = 0;
    size_t i = 0;
    while (i < a->n_tasks) {
        size_t j = i;
        while (j < a->n_tasks && a->tasks[j].type != TASK_BARRIER) j++;
        const long n = (long)(j - i);
        uint32_t work = 0;
        for (long t = 0; t < n; t++) work += a->tasks[i + (size_t)t].type != TASK_SAVE;
        const double t0 = now_seconds();
#ifdef _OPENMP
#pragma omp parallel for collapse(2) schedule(dynamic, 1) num_threads(threads) if (threads > 1)
#endif
        for (long t = 0; t < n; t++)
            for (long blk = 0; blk < blocks; blk++) {
                const task* k = &a->tasks[i + (size_t)t];
                if (k->type == TASK_SAVE) continue;
                run_task_rows(a, k, (uint32_t)blk * rows_per_block, ((uint32_t)blk + 1) * rows_per_block);
            }
        if (work) {
            if (phase < cap) {
                if (phase_seconds) phase_seconds[phase] = now_seconds() - t0;
                if (phase_tasks) phase_tasks[phase] = work;
            }
            phase++;
        }
        i = j + 1;
    }
    if (n_phases) *n_phases = phase;
    a->n_tasks = 0;
    return 0;
}

/* first touch of every atlas page by the threads that will work on it (the allocation is calloc'ed: untouched pages
 * would otherwise be faulted in inside the timed region) */
void orc_atlas_touch(orc_atlas* a, int threads) {
    if (threads < 1) threads = 1;
    for (uint32_t i = 0; i < a->n_att; i++) {
        uint8_t* base = a->att[i].data;
        const long pages = (long)(((size_t)a->atlas_size * orc_tile_bytes(a, i) + 4095) / 4096);
#ifdef _OPENMP
#pragma omp parallel for schedule(static) num_threads(threads) if (threads > 1)
#endif
        for (long pg = 0; pg < pages; pg++) ((volatile uint8_t*)base)[(size_t)pg * 4096] = 0;
    }
}

void orc_split_pixel(uint32_t format, uint32_t T, uint32_t b, orc_coord tile, const float tl[2],
                     const float br[2], const void* src, uint32_t w, uint32_t h, uint32_t px, uint32_t py,
                     const uint32_t prev[4], uint32_t out[4]) {
    vec4 p;
    float n = format == ORC_FORMAT_R16 ? 65535.0f : 255.0f;
    for (int k = 0; k < 4; k++) p.v[k] = (float)prev[k] / n;
    vec4 v = split_pixel_value(format, T, b, tile, tl, br, src, w, h, px, py, p);
    for (int k = 0; k < 4; k++) out[k] = unorm(v.v[k], n);
}

/* ======================================================================== */
/* files: tile_atlas.rs:77-116, 605-612; formats/mod.rs                      */
/* ======================================================================== */

int orc_save_attachment(const orc_atlas* a, uint32_t ai, const char* dir) {
    size_t tb = orc_tile_bytes(a, ai);
    for (uint32_t i = 0; i < a->table_cap; i++) {
        const tile_entry* e = &a->table[i];
        if (!e->used || !e->existing) continue;
        char name[64], path[4096];
        orc_coord_name(e->coord, name, sizeof name);
        snprintf(path, sizeof path, "%s/%s.bin", dir, name);
        FILE* f = fopen(path, "wb");
        if (!f) return -errno;
        size_t wr = fwrite(orc_tile_data(a, ai, e->atlas_index), 1, tb, f);
        fclose(f);
        if (wr != tb) return -5;
    }
    return 0;
}

/* bincode 2 `config::standard()`: little-endian, variable-length ints:
 * u < 251 -> 1 byte; < 2^16 -> 251, u16; < 2^32 -> 252, u32; else 253, u64. */
static size_t put_varint(uint64_t u, uint8_t* out, size_t pos, size_t cap) {
    uint8_t tmp[9];
    size_t n;
    if (u < 251) {
        tmp[0] = (uint8_t)u;
        n = 1;
    } else if (u < (1ull << 16)) {
        tmp[0] = 251;
        tmp[1] = (uint8_t)u;
        tmp[2] = (uint8_t)(u >> 8);
        n = 3;
    } else if (u < (1ull << 32)) {
        tmp[0] = 252;
        for (int k = 0; k < 4; k++) tmp[1 + k] = (uint8_t)(u >> (8 * k));
        n = 5;
    } else {
        tmp[0] = 253;
        for (int k = 0; k < 8; k++) tmp[1 + k] = (uint8_t)(u >> (8 * k));
        n = 9;
    }
    if (out && pos + n <= cap) memcpy(out + pos, tmp, n);
    return pos + n;
}

size_t orc_tc_encode(const orc_coord* tiles, uint32_t n, uint8_t* out, size_t cap) {
    size_t pos = put_varint(n, out, 0, cap); /* Vec length as u64 varint */
    for (uint32_t i = 0; i < n; i++) {
        pos = put_varint(tiles[i].side, out, pos, cap);
        pos = put_varint(tiles[i].lod, out, pos, cap);
        pos = put_varint(tiles[i].x, out, pos, cap);
        pos = put_varint(tiles[i].y, out, pos, cap);
    }
    return pos;
}

static long get_varint(const uint8_t* in, size_t n, size_t* pos, uint64_t* u) {
    if (*pos >= n) return -1;
    uint8_t t = in[(*pos)++];
    int bytes = 0;
    if (t < 251) {
        *u = t;
        return 0;
    } else if (t == 251)
        bytes = 2;
    else if (t == 252)
        bytes = 4;
    else if (t == 253)
        bytes = 8;
    else
        return -1;
    if (*pos + (size_t)bytes > n) return -1;
    uint64_t v = 0;
    for (int k = 0; k < bytes; k++) v |= (uint64_t)in[*pos + (size_t)k] << (8 * k);
    *pos += (size_t)bytes;
    *u = v;
    return 0;
}

long orc_tc_decode(const uint8_t* in, size_t n, orc_coord* tiles, uint32_t cap) {
    size_t pos = 0;
    uint64_t len;
    if (get_varint(in, n, &pos, &len)) return -1;
    for (uint64_t i = 0; i < len; i++) {
        uint64_t v[4];
        for (int k = 0; k < 4; k++)
            if (get_varint(in, n, &pos, &v[k])) return -1;
        if (i < cap) {
            tiles[i].side = (uint32_t)v[0];
            tiles[i].lod = (uint32_t)v[1];
            tiles[i].x = (uint32_t)v[2];
            tiles[i].y = (uint32_t)v[3];
        }
    }
    return (long)len;
}

int orc_save_tile_config(const orc_atlas* a, const char* path) {
    uint32_t n = orc_tile_count(a);
    orc_coord* tiles = (orc_coord*)malloc((n ? n : 1) * sizeof(orc_coord));
    orc_tiles(a, tiles, NULL, n);
    size_t bytes = orc_tc_encode(tiles, n, NULL, 0);
    uint8_t* buf = (uint8_t*)malloc(bytes);
    orc_tc_encode(tiles, n, buf, bytes);
    FILE* f = fopen(path, "wb");
    int rc = 0;
    if (!f)
        rc = -errno;
    else {
        if (fwrite(buf, 1, bytes, f) != bytes) rc = -5;
        fclose(f);
    }
    free(buf);
    free(tiles);
    return rc;
}

/* ======================================================================== */
/* terrain_data/mod.rs:143-219  generate_mipmaps                             */
/* ======================================================================== */

void orc_sample_tile(uint32_t format, uint32_t texture_size, uint32_t border_size, const void* level0,
                     const float atlas_uv[2], float out[4]) {
    const uint32_t T = texture_size, c = T - 2u * border_size;
    const float scale = (float)c / (float)T, offset = (float)border_size / (float)T; /* tile_atlas.rs:183-184 */
    float uv[2], rem[2];
    int ixy[2];
    for (int a = 0; a < 2; a++) {
        const float u = atlas_uv[a] * scale + offset;        /* tile_atlas.rs:255 */
        uv[a] = u * (float)T - 0.5f;                          /* mod.rs:221 */
        rem[a] = fmodf(uv[a], 1.0f);                          /* `uv % 1.0`, mod.rs:223 */
        ixy[a] = (int)uv[a];                                  /* as_ivec2: truncation, mod.rs:224 */
    }
    float v[2][2][4];
    for (int x = 0; x < 2; x++)
        for (int y = 0; y < 2; y++) {                         /* iproduct!(0..2, 0..2), mod.rs:228 */
            int px = ixy[0] + x, py = ixy[1] + y;
            px = px < 0 ? 0 : (px > (int)T - 1 ? (int)T - 1 : px);
            py = py < 0 ? 0 : (py > (int)T - 1 ? (int)T - 1 : py);
            const size_t index = (size_t)py * T + (size_t)px;
            if (format == ORC_FORMAT_R16) {
                v[x][y][0] = (float)((const uint16_t*)level0)[index] / 65535.0f;
                v[x][y][1] = v[x][y][2] = v[x][y][3] = 0.0f;
            } else {
                const uint8_t* t = (const uint8_t*)level0 + 4 * index;
                for (int k = 0; k < 4; k++) v[x][y][k] = (float)t[k] / 255.0f;
            }
        }
    for (int k = 0; k < 4; k++) {                             /* mod.rs:258-262 */
        const float a = v[0][0][k] + (v[0][1][k] - v[0][0][k]) * rem[1];
        const float b = v[1][0][k] + (v[1][1][k] - v[1][0][k]) * rem[1];
        out[k] = a + (b - a) * rem[0];
    }
}

size_t orc_generate_mipmaps(uint32_t format, uint32_t texture_size, uint32_t mip_level_count,
                            const void* level0, void* out) {
    size_t start = 0, parent_size = texture_size, len = (size_t)texture_size * texture_size;
    if (format == ORC_FORMAT_R16) {
        uint16_t* data = (uint16_t*)out;
        memcpy(data, level0, len * 2);
        for (uint32_t mip = 1; mip < mip_level_count; mip++) {
            size_t child_size = parent_size >> 1;
            for (size_t cy = 0; cy < child_size; cy++)
                for (size_t cx = 0; cx < child_size; cx++) {
                    uint32_t value = 0, count = 0;
                    /* iproduct!(0..2, 0..2) -> (x,y): x outer */
                    for (size_t x = 0; x < 2; x++)
                        for (size_t y = 0; y < 2; y++) {
                            size_t index = start + ((cy << 1) + y) * parent_size + (cx << 1) + x;
                            uint32_t d = data[index];
                            if (d != 0) {
                                value += d;
                                count += 1;
                            }
                        }
                    data[len++] = count == 0 ? 0 : (uint16_t)(value / count);
                }
            start += parent_size * parent_size;
            parent_size = child_size;
        }
    } else if (format == ORC_FORMAT_RGBA8) {
        uint8_t* data = (uint8_t*)out;
        memcpy(data, level0, len * 4);
        for (uint32_t mip = 1; mip < mip_level_count; mip++) {
            size_t child_size = parent_size >> 1;
            for (size_t cy = 0; cy < child_size; cy++)
                for (size_t cx = 0; cx < child_size; cx++) {
                    uint64_t value[4] = {0, 0, 0, 0};
                    for (size_t i = 0; i < 4; i++) {
                        size_t px = (cx << 1) + (i >> 1);
                        size_t py = (cy << 1) + (i & 1);
                        size_t index = start + py * parent_size + px;
                        for (int c = 0; c < 4; c++) value[c] += data[index * 4 + c];
                    }
                    for (int c = 0; c < 4; c++) data[len * 4 + c] = (uint8_t)(value[c] / 4);
                    len++;
                }
            start += parent_size * parent_size;
            parent_size = child_size;
        }
    }
    return len;
}

/* ======================================================================== */
/* tiling prepass: shaders/tiling_prepass/*.wgsl + functions.wgsl            */
/* ======================================================================== */

typedef struct {
    uint32_t side, lod;
    uint32_t xy[2];
    float uv[2];
} coordinate; /* types.wgsl:31-40 (vertex/compute variant) */

/* functions.wgsl:164-188.  pow(2.0, f32(d)) is exact for integer d (ldexpf). */
static void coordinate_change_lod(coordinate* c, uint32_t new_lod) {
    int lod_difference = (int)new_lod - (int)c->lod;
    if (lod_difference == 0) return;
    uint32_t delta_count = 1u << (uint32_t)abs(lod_difference);
    float delta_size = ldexpf(1.0f, lod_difference);
    c->lod = new_lod;
    if (lod_difference > 0) {
        for (int k = 0; k < 2; k++) {
            float scaled_uv = c->uv[k] * delta_size;
            c->xy[k] = c->xy[k] * delta_count + (uint32_t)scaled_uv;
            c->uv[k] = scaled_uv - truncf(scaled_uv); /* x % 1.0 = x - 1.0*trunc(x/1.0) */
        }
    } else {
        for (int k = 0; k < 2; k++) {
            uint32_t xy = c->xy[k];
            c->xy[k] = xy / delta_count;
            c->uv[k] = ((float)(xy % delta_count) + c->uv[k]) * delta_size;
        }
    }
}

/* functions.wgsl:133-154 */
static coordinate compute_subdivision_coordinate(const orc_view* v, coordinate c) {
    const orc_side_parameter* params = &v->sides[c.side];
    coordinate vc;
    vc.side = c.side;
    vc.lod = v->origin_lod;
    vc.xy[0] = (uint32_t)params->view_xy[0];
    vc.xy[1] = (uint32_t)params->view_xy[1];
    vc.uv[0] = params->view_uv[0];
    vc.uv[1] = params->view_uv[1];
    coordinate_change_lod(&vc, c.lod);
    int off_x = (int)vc.xy[0] - (int)c.xy[0];
    int off_y = (int)vc.xy[1] - (int)c.xy[1];
    float u = vc.uv[0], w = vc.uv[1];
    if (off_x < 0)
        u = 0.0f;
    else if (off_x > 0)
        u = 1.0f;
    if (off_y < 0)
        w = 0.0f;
    else if (off_y > 0)
        w = 1.0f;
    c.uv[0] = u;
    c.uv[1] = w;
    return c;
}

/* dot/length/normalize: left-to-right sums, one sqrt, component-wise divide */
static float length3(const float a[3]) { return sqrtf(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); }
static void normalize3(float a[3]) {
    float l = length3(a);
    a[0] = a[0] / l;
    a[1] = a[1] / l;
    a[2] = a[2] / l;
}

/* functions.wgsl:73-96 */
static void compute_local_position(const orc_view* v, coordinate c, float out[3]) {
    float tc = (float)(1u << c.lod);
    float u = ((float)c.xy[0] + c.uv[0]) / tc;
    float w = ((float)c.xy[1] + c.uv[1]) / tc;
    if (v->spherical) {
        const float C_SQR = 0.87f * 0.87f; /* functions.wgsl:12 */
        u = (u - 0.5f) / 0.5f;
        w = (w - 0.5f) / 0.5f;
        u = u / sqrtf(1.0f + C_SQR - C_SQR * u * u);
        w = w / sqrtf(1.0f + C_SQR - C_SQR * w * w);
        switch (c.side) {
            case 0: out[0] = -1.0f; out[1] = -w; out[2] = u; break;
            case 1: out[0] = u; out[1] = -w; out[2] = 1.0f; break;
            case 2: out[0] = u; out[1] = 1.0f; out[2] = w; break;
            case 3: out[0] = 1.0f; out[1] = -u; out[2] = w; break;
            case 4: out[0] = w; out[1] = -u; out[2] = -1.0f; break;
            case 5: out[0] = w; out[1] = -1.0f; out[2] = u; break;
            default: out[0] = out[1] = out[2] = 0.0f; break;
        }
        normalize3(out);
    } else {
        out[0] = u - 0.5f;
        out[1] = 0.0f;
        out[2] = w - 0.5f;
    }
}

/* mat3 (column-major) * vec3 : ((c0*x + c1*y) + c2*z) */
static void mat3_mul(const float m[9], const float p[3], float out[3]) {
    for (int r = 0; r < 3; r++) out[r] = m[r] * p[0] + m[3 + r] * p[1] + m[6 + r] * p[2];
}

/* functions.wgsl:117-131 (HIGH_PRECISION is never defined for the prepass, tiling_prepass.rs:61-78) */
static float approximate_view_distance(const orc_view* v, coordinate c) {
    float local_position[3];
    compute_local_position(v, c, local_position);
    /* position_local_to_world (functions.wgsl:26-29): affine * vec4(p, 1) */
    float world_position[3];
    mat3_mul(v->world_from_local, local_position, world_position);
    for (int r = 0; r < 3; r++) world_position[r] = world_position[r] + v->world_from_local[9 + r];
    /* normal_local_to_world (functions.wgsl:14-24) */
    float local_normal[3] = {0.0f, 1.0f, 0.0f};
    if (v->spherical) memcpy(local_normal, local_position, sizeof local_normal);
    float world_normal[3];
    mat3_mul(v->local_from_world_transpose, local_normal, world_normal);
    normalize3(world_normal);
    float d[3];
    for (int r = 0; r < 3; r++)
        d[r] = (world_position[r] + v->approximate_height * world_normal[r]) - v->world_position[r];
    return length3(d);
}

/* refine_tiles.wgsl:17-22 */
int orc_should_be_divided(const orc_view* v, orc_coord tile, float* view_distance) {
    coordinate c;
    c.side = tile.side;
    c.lod = tile.lod;
    c.xy[0] = tile.x;
    c.xy[1] = tile.y;
    c.uv[0] = c.uv[1] = 0.0f;
    c = compute_subdivision_coordinate(v, c);
    float d = approximate_view_distance(v, c);
    if (view_distance) *view_distance = d;
    return d < v->subdivision_distance / (float)(1u << tile.lod);
}

long orc_refine(const orc_view* v, orc_coord* final_tiles, uint32_t cap, uint32_t indirect[4],
                uint32_t* passes_tile_counts) {
    uint32_t N = v->tile_count;
    orc_coord* temporary = (orc_coord*)malloc((size_t)N * sizeof(orc_coord));
    if (!temporary) return -1;
    long overflow = 0;

    /* prepare_root: prepare_prepass.wgsl:4-23 */
    uint32_t tile_count;
    int32_t counter = -1;
    int32_t child_index = (int32_t)(N - 1u);
    int32_t final_index = 0;
    if (v->spherical) {
        tile_count = 6;
        for (uint32_t i = 0; i < 6 && i < N; i++) {
            orc_coord r = {i, 0, 0, 0};
            temporary[i] = r;
        }
    } else {
        tile_count = 1;
        orc_coord r = {0, 0, 0, 0};
        temporary[0] = r;
    }

    for (uint32_t pass = 0; pass <= v->refinement_count; pass++) {
        if (passes_tile_counts) passes_tile_counts[pass] = tile_count;
        /* refine_tiles.wgsl:33-44, invocation ids in increasing order */
        for (uint32_t id = 0; id < tile_count; id++) {
            int32_t pi = (int32_t)(N - 1u) * (counter > 0 ? 1 : 0) - (int32_t)id * counter; /* :9-11 */
            orc_coord tile = temporary[pi];
            if (orc_should_be_divided(v, tile, NULL)) {
                for (uint32_t i = 0; i < 4; i++) { /* subdivide :24-31 */
                    orc_coord child = {tile.side, tile.lod + 1u, (tile.x << 1) + (i & 1u),
                                       (tile.y << 1) + ((i >> 1) & 1u)};
                    int32_t ci = child_index;
                    child_index += counter;
                    if (ci < 0 || ci >= (int32_t)N)
                        overflow = 1;
                    else
                        temporary[ci] = child;
                }
            } else {
                int32_t fi = final_index++;
                if ((uint32_t)fi < cap)
                    final_tiles[fi] = tile;
                else
                    overflow = 1;
            }
        }
        if (pass == v->refinement_count) break;
        /* prepare_next: prepare_prepass.wgsl:25-36 */
        if (counter == 1) {
            tile_count = (uint32_t)child_index;
            child_index = (int32_t)(N - 1u);
        } else {
            tile_count = N - 1u - (uint32_t)child_index;
            child_index = 0;
        }
        counter = -counter;
    }
    /* prepare_render: prepare_prepass.wgsl:38-44 */
    if (indirect) {
        indirect[0] = v->vertices_per_tile * (uint32_t)final_index;
        indirect[1] = 1;
        indirect[2] = 0;
        indirect[3] = 0;
    }
    free(temporary);
    return overflow ? -1 : (long)final_index;
}
