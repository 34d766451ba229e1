/*
 * bt_oracle_tree.c — CPU ORACLE, part 2.  TEST INFRASTRUCTURE ONLY (see bt_oracle.h).
 *
 * Restates, function by function and in f64 like the reference, the CPU side of the per-frame node selection:
 *   math/terrain_model.rs   TerrainModel transforms, scale()
 *   math/coordinate.rs      Coordinate::{from_world_position, world_position, project_to_side}
 *   math/ellipsoid.rs       project_point_ellipsoid
 *   terrain_data/tile_tree.rs  TileTree::{new, compute_tree_xy, compute_origin, compute_tile_distance, compute_blend,
 *                              lookup_tile, update, adjust_to_tile_atlas}
 *   terrain_data/tile_atlas.rs TileAtlasState::{allocate_tile, request_tile, release_tile, get_best_tile,
 *                              loaded_tile_attachment}
 *   terrain_data/mod.rs:265-307  sample_attachment / sample_height
 *   render/terrain_view_bind_group.rs:98-116 + math/terrain_model.rs:262-290  the prepass' view uniforms
 *
 * PARITY UNPINNED (no reference tests; the reference cannot be built here).  Definitions where the reference defers
 * to a library: glam's DMat4::from_scale_rotation_translation(scale, IDENTITY, t) is applied as scale * p + t and its
 * general `.inverse()` as (p - t) / scale; `.powf(0.5)` is sqrt; DVec3::normalize() = v * (1.0 / length);
 * DVec2::fract() = v - trunc(v) (glam 0.27).  Rust's float -> int `as` casts saturate.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "bt_oracle.h"

#define C_SQR (0.87 * 0.87) /* math/mod.rs:13 */

typedef struct {
    double x, y, z;
} dvec3;
typedef struct {
    double x, y;
} dvec2;

static double dot(dvec3 a, dvec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static double length(dvec3 a) { return sqrt(dot(a, a)); }
static dvec3 normalize(dvec3 a) {
    double r = 1.0 / length(a);
    dvec3 o = {a.x * r, a.y * r, a.z * r};
    return o;
}
static dvec3 sub(dvec3 a, dvec3 b) {
    dvec3 o = {a.x - b.x, a.y - b.y, a.z - b.z};
    return o;
}

/* ---- TerrainModel ---------------------------------------------------------------------------------- */
static int is_spherical(const orc_model* m) { return m->kind != 0; } /* terrain_model.rs:54-60 */
static uint32_t side_count(const orc_model* m) { return is_spherical(m) ? 6u : 1u; }
static dvec3 model_scale_vec(const orc_model* m) {
    dvec3 s = {m->a, m->kind == 2 ? m->b : m->a, m->a}; /* :96-138 */
    return s;
}
static double model_scale(const orc_model* m) { /* :183-193 */
    if (m->kind == 0) return m->a / 2.0;
    if (m->kind == 1) return m->a;
    return (m->a + m->b) / 2.0;
}
static dvec3 world_from_local_point(const orc_model* m, dvec3 p) {
    dvec3 s = model_scale_vec(m);
    dvec3 o = {s.x * p.x + m->position[0], s.y * p.y + m->position[1], s.z * p.z + m->position[2]};
    return o;
}
static dvec3 world_from_local_vector(const orc_model* m, dvec3 v) {
    dvec3 s = model_scale_vec(m);
    dvec3 o = {s.x * v.x, s.y * v.y, s.z * v.z};
    return o;
}
static dvec3 local_from_world_point(const orc_model* m, dvec3 p) {
    dvec3 s = model_scale_vec(m);
    dvec3 o = {(p.x - m->position[0]) / s.x, (p.y - m->position[1]) / s.y, (p.z - m->position[2]) / s.z};
    return o;
}
/* :140-152 */
static dvec3 position_local_to_world(const orc_model* m, dvec3 local_position, double height) {
    dvec3 world_position = world_from_local_point(m, local_position);
    dvec3 up = {0.0, 1.0, 0.0};
    dvec3 world_normal = normalize(world_from_local_vector(m, is_spherical(m) ? local_position : up));
    dvec3 o = {world_position.x + height * world_normal.x, world_position.y + height * world_normal.y,
               world_position.z + height * world_normal.z};
    return o;
}

/* ---- math/ellipsoid.rs -------------------------------------------------------------------------------- */
#define MAX_ITERATIONS 1074
static double get_root_2d(dvec2 r, dvec2 z, double g) { /* :117-142 */
    dvec2 n = {r.x * z.x, r.y * z.y};
    double s0 = z.y - 1.0;
    double s1 = g < 0.0 ? 0.0 : sqrt(n.x * n.x + n.y * n.y) - 1.0;
    double s = 0.0;
    for (int i = 0; i < MAX_ITERATIONS; i++) {
        s = (s0 + s1) / 2.0;
        if (s == s0 || s == s1) break;
        dvec2 ratio = {n.x / (s + r.x), n.y / (s + r.y)};
        double gg = ratio.x * ratio.x + ratio.y * ratio.y - 1.0;
        if (gg < 0.0)
            s1 = s;
        else if (gg == 0.0)
            break;
        else
            s0 = s;
    }
    return s;
}
static double get_root_3d(dvec3 r, dvec3 z, double g) { /* :90-115 */
    dvec3 n = {r.x * z.x, r.y * z.y, r.z * z.z};
    double s0 = z.z - 1.0;
    double s1 = g < 0.0 ? 0.0 : length(n) - 1.0;
    double s = 0.0;
    for (int i = 0; i < MAX_ITERATIONS; i++) {
        s = (s0 + s1) / 2.0;
        if (s == s0 || s == s1) break;
        dvec3 ratio = {n.x / (s + r.x), n.y / (s + r.y), n.z / (s + r.z)};
        double gg = dot(ratio, ratio) - 1.0;
        if (gg < 0.0)
            s1 = s;
        else if (gg == 0.0)
            break;
        else
            s0 = s;
    }
    return s;
}
static dvec2 project_point_ellipse(dvec2 e, dvec2 y) { /* :64-88 */
    dvec2 o;
    if (y.y > 0.0) {
        if (y.x > 0.0) {
            dvec2 z = {y.x / e.x, y.y / e.y};
            double g = z.x * z.x + z.y * z.y - 1.0;
            if (g != 0.0) {
                dvec2 r = {(e.x * e.x) / (e.y * e.y), 1.0};
                double root = get_root_2d(r, z, g);
                o.x = r.x * y.x / (root + r.x);
                o.y = r.y * y.y / (root + r.y);
                return o;
            }
            return y;
        }
        o.x = 0.0;
        o.y = e.y;
        return o;
    }
    double numer0 = e.x * y.x, denom0 = e.x * e.x - e.y * e.y;
    if (numer0 < denom0) {
        double xde0 = numer0 / denom0;
        o.x = e.x * xde0;
        o.y = e.y * sqrt(1.0 - xde0 * xde0);
        return o;
    }
    o.x = e.x;
    o.y = 0.0;
    return o;
}
static double signum(double v) { return isnan(v) ? v : (signbit(v) ? -1.0 : 1.0); }
void orc_project_point_ellipsoid(const double ev[3], const double yv[3], double out[3]) { /* :12-62 */
    dvec3 e = {ev[0], ev[1], ev[2]};
    dvec3 sign = {signum(yv[0]), signum(yv[1]), signum(yv[2])};
    dvec3 y = {fabs(yv[0]), fabs(yv[2]), fabs(yv[1])}; /* y.xzy().abs() */
    dvec3 x;
    if (y.z > 0.0) {
        if (y.y > 0.0) {
            if (y.x > 0.0) {
                dvec3 z = {y.x / e.x, y.y / e.y, y.z / e.z};
                double g = dot(z, z) - 1.0;
                if (g != 0.0) {
                    dvec3 r = {(e.x * e.x) / (e.z * e.z), (e.y * e.y) / (e.z * e.z), 1.0};
                    double root = get_root_3d(r, z, g);
                    x.x = r.x * y.x / (root + r.x);
                    x.y = r.y * y.y / (root + r.y);
                    x.z = r.z * y.z / (root + r.z);
                } else {
                    x = y;
                }
            } else {
                dvec2 ee = {e.y, e.z}, yy = {y.y, y.z};
                dvec2 p = project_point_ellipse(ee, yy); /* .extend(0.0).zxy() */
                x.x = 0.0;
                x.y = p.x;
                x.z = p.y;
            }
        } else {
            if (y.x > 0.0) {
                dvec2 ee = {e.x, e.z}, yy = {y.x, y.z};
                dvec2 p = project_point_ellipse(ee, yy); /* .extend(0.0).xzy() */
                x.x = p.x;
                x.y = 0.0;
                x.z = p.y;
            } else {
                x.x = 0.0;
                x.y = 0.0;
                x.z = e.z;
            }
        }
    } else {
        double denom0 = e.x * e.x - e.z * e.z, denom1 = e.y * e.y - e.z * e.z;
        double numer0 = e.x * y.x, numer1 = e.y * y.y;
        int found = 0;
        if (numer0 < denom0 && numer1 < denom1) {
            double xde0 = numer0 / denom0, xde1 = numer1 / denom1;
            double xde0sqr = xde0 * xde0, xde1sqr = xde1 * xde1;
            double discr = 1.0 - xde0sqr - xde1sqr;
            if (discr > 0.0) {
                x.x = e.x * xde0;
                x.y = e.y * xde1;
                x.z = e.z * sqrt(discr);
                found = 1;
            }
        }
        if (!found) {
            dvec2 ee = {e.x, e.y}, yy = {y.x, y.y};
            dvec2 p = project_point_ellipse(ee, yy);
            x.x = p.x;
            x.y = p.y;
            x.z = 0.0;
        }
    }
    out[0] = sign.x * x.x; /* sign * x.xzy() */
    out[1] = sign.y * x.z;
    out[2] = sign.z * x.y;
}

/* terrain_model.rs:154-176 */
static dvec3 position_world_to_local(const orc_model* m, dvec3 world_position) {
    if (m->kind == 0) {
        dvec3 p = local_from_world_point(m, world_position);
        dvec3 o = {1.0 * p.x, 0.0 * p.y, 1.0 * p.z};
        return o;
    }
    if (m->kind == 1) return normalize(local_from_world_point(m, world_position));
    /* ellipsoid_from_world = inverse(from_rotation_translation(IDENTITY, position)): p - position */
    double ellipsoid_position[3] = {world_position.x - m->position[0], world_position.y - m->position[1],
                                    world_position.z - m->position[2]};
    double e[3] = {m->a, m->a, m->b}, surface[3];
    orc_project_point_ellipsoid(e, ellipsoid_position, surface);
    dvec3 sp = {surface[0], surface[1], surface[2]};
    return normalize(local_from_world_point(m, sp));
}
static dvec3 surface_position(const orc_model* m, dvec3 world_position, double height) { /* :178-180 */
    return position_local_to_world(m, position_world_to_local(m, world_position), height);
}

/* ---- Coordinate ------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t side;
    dvec2 uv;
} coordinate;

static double clampd(double v, double lo, double hi) { return v < lo ? lo : (v > hi ? hi : v); }

static coordinate coordinate_from_world_position(dvec3 world_position, const orc_model* m) { /* coordinate.rs:69-113 */
    dvec3 local_position = position_world_to_local(m, world_position);
    coordinate c;
    if (is_spherical(m)) {
        dvec3 normal = local_position;
        dvec3 a = {fabs(normal.x), fabs(normal.y), fabs(normal.z)};
        dvec2 uv;
        if (a.x > a.y && a.x > a.z) {
            if (normal.x < 0.0) {
                c.side = 0;
                uv.x = -normal.z / normal.x;
                uv.y = normal.y / normal.x;
            } else {
                c.side = 3;
                uv.x = -normal.y / normal.x;
                uv.y = normal.z / normal.x;
            }
        } else if (a.z > a.y) {
            if (normal.z > 0.0) {
                c.side = 1;
                uv.x = normal.x / normal.z;
                uv.y = -normal.y / normal.z;
            } else {
                c.side = 4;
                uv.x = normal.y / normal.z;
                uv.y = -normal.x / normal.z;
            }
        } else {
            if (normal.y > 0.0) {
                c.side = 2;
                uv.x = normal.x / normal.y;
                uv.y = normal.z / normal.y;
            } else {
                c.side = 5;
                uv.x = -normal.z / normal.y;
                uv.y = -normal.x / normal.y;
            }
        }
        dvec2 w = {uv.x * sqrt((1.0 + C_SQR) / (1.0 + C_SQR * uv.x * uv.x)),
                   uv.y * sqrt((1.0 + C_SQR) / (1.0 + C_SQR * uv.y * uv.y))};
        c.uv.x = 0.5 * w.x + 0.5;
        c.uv.y = 0.5 * w.y + 0.5;
    } else {
        c.side = 0;
        c.uv.x = clampd(local_position.x + 0.5, 0.0, 1.0);
        c.uv.y = clampd(local_position.z + 0.5, 0.0, 1.0);
    }
    return c;
}

static dvec3 coordinate_world_position(coordinate c, const orc_model* m, float height) { /* :115-135 */
    dvec3 local_position;
    if (is_spherical(m)) {
        dvec2 w = {(c.uv.x - 0.5) / 0.5, (c.uv.y - 0.5) / 0.5};
        dvec2 uv = {w.x / sqrt(1.0 + C_SQR - C_SQR * w.x * w.x), w.y / sqrt(1.0 + C_SQR - C_SQR * w.y * w.y)};
        dvec3 v;
        switch (c.side) {
            case 0: v.x = -1.0; v.y = -uv.y; v.z = uv.x; break;
            case 1: v.x = uv.x; v.y = -uv.y; v.z = 1.0; break;
            case 2: v.x = uv.x; v.y = 1.0; v.z = uv.y; break;
            case 3: v.x = 1.0; v.y = -uv.x; v.z = uv.y; break;
            case 4: v.x = uv.y; v.y = -uv.x; v.z = -1.0; break;
            default: v.x = uv.y; v.y = -1.0; v.z = uv.x; break;
        }
        local_position = normalize(v);
    } else {
        local_position.x = c.uv.x - 0.5;
        local_position.y = 0.0;
        local_position.z = c.uv.y - 0.5;
    }
    return position_local_to_world(m, local_position, (double)height);
}

/* SideInfo (coordinate.rs:18-53): 0 Fixed0, 1 Fixed1, 2 PositiveS, 3 PositiveT */
static const int EVEN_LIST[6][2] = {{2, 3}, {0, 3}, {0, 2}, {3, 2}, {3, 0}, {2, 0}};
static const int ODD_LIST[6][2] = {{2, 3}, {2, 1}, {3, 1}, {3, 2}, {1, 2}, {1, 3}};

static coordinate coordinate_project_to_side(coordinate c, uint32_t side, const orc_model* m) { /* :137-151 */
    if (!is_spherical(m)) return c;
    const int* info = (c.side % 2 == 0 ? EVEN_LIST : ODD_LIST)[(6 + side - c.side) % 6];
    double uv[2];
    for (int k = 0; k < 2; k++)
        uv[k] = info[k] == 0 ? 0.0 : info[k] == 1 ? 1.0 : info[k] == 2 ? c.uv.x : c.uv.y;
    coordinate o;
    o.side = side;
    o.uv.x = uv[0];
    o.uv.y = uv[1];
    return o;
}

uint32_t orc_coordinate_from_world_position(const orc_model* model, const double world[3], double uv[2]) {
    dvec3 w = {world[0], world[1], world[2]};
    coordinate c = coordinate_from_world_position(w, model);
    uv[0] = c.uv.x;
    uv[1] = c.uv.y;
    return c.side;
}
void orc_coordinate_world_position(const orc_model* model, uint32_t side, const double uv[2], float height, double out[3]) {
    coordinate c;
    c.side = side;
    c.uv.x = uv[0];
    c.uv.y = uv[1];
    dvec3 p = coordinate_world_position(c, model, height);
    out[0] = p.x;
    out[1] = p.y;
    out[2] = p.z;
}

/* Rust `as` casts */
static uint32_t as_u32(double v) { return !(v > 0.0) ? 0u : (v >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)v); }
static int32_t as_i32(double v) {
    return isnan(v) ? 0 : (v <= -2147483648.0 ? INT32_MIN : (v >= 2147483647.0 ? INT32_MAX : (int32_t)v));
}

/* ---- the prepass' view uniforms ------------------------------------------------------------------- */
void orc_view_state_from_config(const orc_model* model, const orc_view_config* vc, const double view_world_position[3],
                                float approximate_height, orc_view* out) {
    memset(out, 0, sizeof *out);
    out->spherical = (uint32_t)is_spherical(model);
    out->tile_count = vc->geometry_tile_count;
    out->refinement_count = vc->refinement_count;
    out->vertices_per_tile = 2u * vc->grid_size * (vc->grid_size + 2u); /* terrain_view_bind_group.rs:106 */
    /* tile_tree.rs:148-150, then `as f32` (terrain_view_bind_group.rs:111) */
    double subdivision_distance = vc->morph_distance * model_scale(model) * (1.0 + vc->subdivision_tolerance);
    out->subdivision_distance = (float)subdivision_distance;
    out->origin_lod = vc->origin_lod;
    out->approximate_height = approximate_height;
    /* terrain_model.rs:262-290 */
    double origin_count = (double)(1u << vc->origin_lod);
    dvec3 view = {view_world_position[0], view_world_position[1], view_world_position[2]};
    coordinate view_coordinate = coordinate_from_world_position(view, model);
    for (uint32_t side = 0; side < 6; side++) {
        coordinate c = coordinate_project_to_side(view_coordinate, side, model);
        double sx = c.uv.x * origin_count, sy = c.uv.y * origin_count;
        out->sides[side].view_xy[0] = as_i32(sx);
        out->sides[side].view_xy[1] = as_i32(sy);
        out->sides[side].view_uv[0] = (float)(sx - trunc(sx));
        out->sides[side].view_uv[1] = (float)(sy - trunc(sy));
    }
    for (int i = 0; i < 3; i++) out->world_position[i] = (float)view_world_position[i]; /* culling_bind_group.rs:50 */
    /* TerrainModel::transform() (terrain_model.rs:195-201) -> Bevy's MeshUniform: world_from_local columns +
     * translation, local_from_world_transpose = diag(1 / scale) in f32 */
    dvec3 s = model_scale_vec(model);
    float sc[3] = {(float)s.x, (float)s.y, (float)s.z};
    for (int c = 0; c < 3; c++) {
        out->world_from_local[3 * c + c] = sc[c];
        out->world_from_local[9 + c] = (float)model->position[c];
        out->local_from_world_transpose[3 * c + c] = 1.0f / sc[c];
    }
}

/* ---- streaming TileAtlasState ------------------------------------------------------------------------ */
typedef struct {
    orc_coord coordinate;
    uint32_t atlas_index;
    uint32_t requests;
    uint32_t loading; /* 0 = Loaded, n = Loading(n) */
    int used;
} stream_tile;

typedef struct {
    orc_coord coordinate;
    uint32_t atlas_index, attachment_index;
} stream_load;

struct orc_stream {
    uint32_t atlas_size, attachment_count;
    stream_tile* tile_states; /* at most atlas_size live entries (linear search: test sizes) */
    uint32_t tile_state_cap;
    orc_atlas_tile* unused; /* VecDeque<AtlasTile> as an array with head */
    uint32_t unused_len;
    orc_coord* existing;
    uint32_t existing_len, existing_cap;
    stream_load* to_load;
    uint32_t to_load_len, to_load_cap;
};

static int coord_equal(orc_coord a, orc_coord b) { return a.side == b.side && a.lod == b.lod && a.x == b.x && a.y == b.y; }
static const orc_coord INVALID_COORD = {ORC_INVALID, ORC_INVALID, ORC_INVALID, ORC_INVALID};

orc_stream* orc_stream_new(uint32_t atlas_size, uint32_t attachment_count) { /* TileAtlasState::new :302-325 */
    orc_stream* s = (orc_stream*)calloc(1, sizeof *s);
    s->atlas_size = atlas_size;
    s->attachment_count = attachment_count;
    s->tile_state_cap = atlas_size + 1;
    s->tile_states = (stream_tile*)calloc(s->tile_state_cap, sizeof(stream_tile));
    s->unused = (orc_atlas_tile*)calloc(atlas_size + 1, sizeof(orc_atlas_tile));
    for (uint32_t i = 0; i < atlas_size; i++) {
        s->unused[i].coordinate = INVALID_COORD;
        s->unused[i].atlas_index = i;
    }
    s->unused_len = atlas_size;
    return s;
}
void orc_stream_free(orc_stream* s) {
    if (!s) return;
    free(s->tile_states);
    free(s->unused);
    free(s->existing);
    free(s->to_load);
    free(s);
}
void orc_stream_add_existing(orc_stream* s, const orc_coord* tiles, uint32_t n) {
    for (uint32_t i = 0; i < n; i++) {
        if (s->existing_len == s->existing_cap) {
            s->existing_cap = s->existing_cap ? 2 * s->existing_cap : 256;
            s->existing = (orc_coord*)realloc(s->existing, s->existing_cap * sizeof(orc_coord));
        }
        s->existing[s->existing_len++] = tiles[i];
    }
}
static int stream_exists(const orc_stream* s, orc_coord c) {
    for (uint32_t i = 0; i < s->existing_len; i++)
        if (coord_equal(s->existing[i], c)) return 1;
    return 0;
}
static stream_tile* stream_find(const orc_stream* s, orc_coord c) {
    for (uint32_t i = 0; i < s->tile_state_cap; i++)
        if (s->tile_states[i].used && coord_equal(s->tile_states[i].coordinate, c)) return &s->tile_states[i];
    return NULL;
}
static void stream_remove(orc_stream* s, orc_coord c) {
    stream_tile* t = stream_find(s, c);
    if (t) t->used = 0;
}
static stream_tile* stream_insert(orc_stream* s, orc_coord c) {
    for (uint32_t i = 0; i < s->tile_state_cap; i++)
        if (!s->tile_states[i].used) {
            s->tile_states[i].used = 1;
            s->tile_states[i].coordinate = c;
            return &s->tile_states[i];
        }
    return NULL;
}
/* allocate_tile :383-389 */
static int stream_allocate(orc_stream* s, uint32_t* atlas_index) {
    if (s->unused_len == 0) return -2; /* "Atlas out of indices" */
    orc_atlas_tile unused_tile = s->unused[0];
    memmove(s->unused, s->unused + 1, (size_t)(s->unused_len - 1) * sizeof(orc_atlas_tile));
    s->unused_len--;
    stream_remove(s, unused_tile.coordinate);
    *atlas_index = unused_tile.atlas_index;
    return 0;
}
int orc_stream_request_tile(orc_stream* s, orc_coord c) { /* :418-457 */
    if (!stream_exists(s, c)) return 0;
    stream_tile* tile = stream_find(s, c);
    if (tile) {
        if (tile->requests == 0) { /* the tile is now used again: unused_tiles.retain(..) */
            uint32_t k = 0;
            for (uint32_t i = 0; i < s->unused_len; i++)
                if (s->unused[i].atlas_index != tile->atlas_index) s->unused[k++] = s->unused[i];
            s->unused_len = k;
        }
        tile->requests += 1;
        return 0;
    }
    uint32_t atlas_index;
    int rc = stream_allocate(s, &atlas_index);
    if (rc) return rc;
    tile = stream_insert(s, c);
    tile->requests = 1;
    tile->loading = s->attachment_count;
    tile->atlas_index = atlas_index;
    for (uint32_t ai = 0; ai < s->attachment_count; ai++) {
        if (s->to_load_len == s->to_load_cap) {
            s->to_load_cap = s->to_load_cap ? 2 * s->to_load_cap : 256;
            s->to_load = (stream_load*)realloc(s->to_load, s->to_load_cap * sizeof(stream_load));
        }
        stream_load l = {c, atlas_index, ai};
        s->to_load[s->to_load_len++] = l;
    }
    return 0;
}
int orc_stream_release_tile(orc_stream* s, orc_coord c) { /* :459-476 */
    if (!stream_exists(s, c)) return 0;
    stream_tile* tile = stream_find(s, c);
    if (!tile || tile->requests == 0) return -1; /* "Tried releasing a tile, which is not present." */
    tile->requests -= 1;
    if (tile->requests == 0) {
        s->unused[s->unused_len].coordinate = c;
        s->unused[s->unused_len].atlas_index = tile->atlas_index;
        s->unused_len++;
    }
    return 0;
}
orc_tree_entry orc_stream_get_best_tile(const orc_stream* s, orc_coord c) { /* :478-503 */
    orc_coord best = c;
    orc_tree_entry none = {ORC_INVALID, ORC_INVALID};
    for (;;) {
        if (coord_equal(best, INVALID_COORD) || best.lod == ORC_INVALID) return none;
        stream_tile* t = stream_find(s, best);
        if (t && t->loading == 0) {
            orc_tree_entry e = {t->atlas_index, best.lod};
            return e;
        }
        best = orc_parent(best);
    }
}
uint32_t orc_stream_pending_loads(const orc_stream* s) { return s->to_load_len; }
uint32_t orc_stream_finish_loads(orc_stream* s, uint32_t n, uint32_t* out) { /* :327-359 */
    uint32_t done = 0;
    if (n > s->to_load_len) n = s->to_load_len;
    for (uint32_t i = 0; i < n; i++) {
        stream_load l = s->to_load[i];
        stream_tile* t = stream_find(s, l.coordinate);
        if (t && t->atlas_index == l.atlas_index && t->loading > 0) t->loading -= 1; /* Loading(1) -> Loaded */
        if (out) {
            out[5 * done + 0] = l.coordinate.side;
            out[5 * done + 1] = l.coordinate.lod;
            out[5 * done + 2] = l.coordinate.x;
            out[5 * done + 3] = l.coordinate.y;
            out[5 * done + 4] = l.atlas_index;
        }
        done++;
    }
    memmove(s->to_load, s->to_load + n, (size_t)(s->to_load_len - n) * sizeof(stream_load));
    s->to_load_len -= n;
    return done;
}
uint32_t orc_stream_atlas_index(const orc_stream* s, orc_coord c) {
    stream_tile* t = stream_find(s, c);
    return t ? t->atlas_index : ORC_INVALID;
}

/* ---- TileTree ---------------------------------------------------------------------------------------- */
typedef struct {
    orc_coord coordinate;
    int requested;
} tree_tile;

struct orc_tile_tree {
    orc_model model;
    uint32_t lod_count, tree_size, sides;
    double morph_distance, blend_distance, load_distance, subdivision_distance;
    float blend_range;
    dvec3 view_world_position;
    float approximate_height;
    uint32_t* origins;       /* [side][lod][2] */
    orc_tree_entry* data;    /* [side][lod][x][y] */
    tree_tile* tiles;        /* [side][lod][x][y] */
    orc_coord *released, *requested;
    uint32_t released_len, requested_len, list_cap;
};

orc_tile_tree* orc_tile_tree_new(const orc_model* model, uint32_t lod_count, const orc_view_config* vc) { /* :135-173 */
    orc_tile_tree* t = (orc_tile_tree*)calloc(1, sizeof *t);
    double scale = model_scale(model);
    t->model = *model;
    t->lod_count = lod_count;
    t->tree_size = vc->tree_size;
    t->sides = side_count(model);
    t->morph_distance = vc->morph_distance * scale;
    t->blend_distance = vc->blend_distance * scale;
    t->load_distance = vc->load_distance * scale;
    t->subdivision_distance = vc->morph_distance * scale * (1.0 + vc->subdivision_tolerance);
    t->blend_range = vc->blend_range;
    t->approximate_height = (model->min_height + model->max_height) / 2.0f;
    size_t n = (size_t)t->sides * lod_count * vc->tree_size * vc->tree_size;
    t->origins = (uint32_t*)calloc((size_t)t->sides * lod_count * 2, sizeof(uint32_t));
    t->data = (orc_tree_entry*)malloc(n * sizeof(orc_tree_entry));
    t->tiles = (tree_tile*)malloc(n * sizeof(tree_tile));
    for (size_t i = 0; i < n; i++) {
        t->data[i].atlas_index = ORC_INVALID;
        t->data[i].atlas_lod = ORC_INVALID;
        t->tiles[i].coordinate = INVALID_COORD;
        t->tiles[i].requested = 0;
    }
    t->list_cap = (uint32_t)(2 * n + 16);
    t->released = (orc_coord*)malloc(t->list_cap * sizeof(orc_coord));
    t->requested = (orc_coord*)malloc(t->list_cap * sizeof(orc_coord));
    return t;
}
void orc_tile_tree_free(orc_tile_tree* t) {
    if (!t) return;
    free(t->origins);
    free(t->data);
    free(t->tiles);
    free(t->released);
    free(t->requested);
    free(t);
}
uint32_t orc_tile_tree_node_count(const orc_tile_tree* t) { return t->sides * t->lod_count * t->tree_size * t->tree_size; }
void orc_tile_tree_set_approximate_height(orc_tile_tree* t, float h) { t->approximate_height = h; }

static dvec2 compute_tree_xy(coordinate c, double tile_count) { /* :175-178 */
    dvec2 v = {c.uv.x * tile_count, c.uv.y * tile_count};
    double cap = tile_count - 0.000001;
    v.x = fmin(v.x, cap);
    v.y = fmin(v.y, cap);
    return v;
}
static void compute_origin(const orc_tile_tree* t, coordinate c, uint32_t lod, uint32_t out[2]) { /* :180-192 */
    double tile_count = (double)(1u << lod);
    dvec2 tree_xy = compute_tree_xy(c, tile_count);
    double v[2] = {round(tree_xy.x - 0.5 * (double)t->tree_size), round(tree_xy.y - 0.5 * (double)t->tree_size)};
    for (int k = 0; k < 2; k++) { /* DVec2::clamp = max(min).min(max) */
        v[k] = fmax(v[k], 0.0);
        v[k] = fmin(v[k], tile_count - (double)t->tree_size);
        out[k] = as_u32(v[k]);
    }
}
static double compute_tile_distance(const orc_tile_tree* t, orc_coord tile, coordinate view_coordinate) { /* :194-221 */
    double tile_count = (double)(1u << tile.lod);
    int32_t tile_x = (int32_t)tile.x, tile_y = (int32_t)tile.y;
    dvec2 view_tile_xy = compute_tree_xy(view_coordinate, tile_count);
    int32_t off_x = as_i32(view_tile_xy.x) - tile_x, off_y = as_i32(view_tile_xy.y) - tile_y;
    dvec2 offset = {fmod(view_tile_xy.x, 1.0), fmod(view_tile_xy.y, 1.0)};
    if (off_x < 0)
        offset.x = 0.0;
    else if (off_x > 0)
        offset.x = 1.0;
    if (off_y < 0)
        offset.y = 0.0;
    else if (off_y > 0)
        offset.y = 1.0;
    coordinate c;
    c.side = tile.side;
    c.uv.x = ((double)tile_x + offset.x) / tile_count;
    c.uv.y = ((double)tile_y + offset.y) / tile_count;
    dvec3 tile_world_position = coordinate_world_position(c, &t->model, t->approximate_height);
    return length(sub(tile_world_position, t->view_world_position));
}

static size_t slot_index(const orc_tile_tree* t, uint32_t side, uint32_t lod, uint32_t x, uint32_t y) {
    return (((size_t)side * t->lod_count + lod) * t->tree_size + x) * t->tree_size + y;
}

void orc_tile_tree_update(orc_tile_tree* t, const double view_position[3]) { /* :268-333 */
    t->view_world_position.x = view_position[0];
    t->view_world_position.y = view_position[1];
    t->view_world_position.z = view_position[2];
    t->released_len = t->requested_len = 0;
    coordinate view_coordinate0 = coordinate_from_world_position(t->view_world_position, &t->model);
    for (uint32_t side = 0; side < t->sides; side++) {
        coordinate view_coordinate = coordinate_project_to_side(view_coordinate0, side, &t->model);
        for (uint32_t lod = 0; lod < t->lod_count; lod++) {
            uint32_t origin[2];
            compute_origin(t, view_coordinate, lod, origin);
            t->origins[(side * t->lod_count + lod) * 2 + 0] = origin[0];
            t->origins[(side * t->lod_count + lod) * 2 + 1] = origin[1];
            for (uint32_t x = 0; x < t->tree_size; x++)
                for (uint32_t y = 0; y < t->tree_size; y++) { /* iproduct!(0..n, 0..n): x outer */
                    orc_coord tile_coordinate = {side, lod, origin[0] + x, origin[1] + y};
                    double tile_distance = compute_tile_distance(t, tile_coordinate, view_coordinate);
                    double load_distance = t->load_distance / (double)(1u << tile_coordinate.lod);
                    int state = (lod == 0 || tile_distance < load_distance) ? 1 : 0;
                    tree_tile* tile = &t->tiles[slot_index(t, side, lod, tile_coordinate.x % t->tree_size,
                                                           tile_coordinate.y % t->tree_size)];
                    if (!coord_equal(tile_coordinate, tile->coordinate)) {
                        if (tile->requested) {
                            tile->requested = 0;
                            t->released[t->released_len++] = tile->coordinate;
                        }
                        tile->coordinate = tile_coordinate;
                    }
                    if (!tile->requested && state) {
                        tile->requested = 1;
                        t->requested[t->requested_len++] = tile->coordinate;
                    } else if (tile->requested && !state) {
                        tile->requested = 0;
                        t->released[t->released_len++] = tile->coordinate;
                    }
                }
        }
    }
}
uint32_t orc_tile_tree_released(const orc_tile_tree* t, orc_coord* out, uint32_t cap) {
    for (uint32_t i = 0; i < t->released_len && i < cap; i++) out[i] = t->released[i];
    return t->released_len;
}
uint32_t orc_tile_tree_requested(const orc_tile_tree* t, orc_coord* out, uint32_t cap) {
    for (uint32_t i = 0; i < t->requested_len && i < cap; i++) out[i] = t->requested[i];
    return t->requested_len;
}
int orc_tile_tree_apply_requests(orc_tile_tree* t, orc_stream* s) { /* tile_atlas.rs:590-600 */
    for (uint32_t i = 0; i < t->released_len; i++) {
        int rc = orc_stream_release_tile(s, t->released[i]);
        if (rc) return rc;
    }
    t->released_len = 0;
    for (uint32_t i = 0; i < t->requested_len; i++) {
        int rc = orc_stream_request_tile(s, t->requested[i]);
        if (rc) return rc;
    }
    t->requested_len = 0;
    return 0;
}
void orc_tile_tree_adjust_to_tile_atlas(orc_tile_tree* t, const orc_stream* s) { /* :363-374 */
    uint32_t n = orc_tile_tree_node_count(t);
    for (uint32_t i = 0; i < n; i++) t->data[i] = orc_stream_get_best_tile(s, t->tiles[i].coordinate);
}
void orc_tile_tree_read(const orc_tile_tree* t, orc_tree_entry* entries, uint32_t* origins, orc_coord* nodes, uint32_t* requested) {
    uint32_t n = orc_tile_tree_node_count(t);
    for (uint32_t i = 0; i < n; i++) {
        if (entries) entries[i] = t->data[i];
        if (nodes) nodes[i] = t->tiles[i].coordinate;
        if (requested) requested[i] = (uint32_t)t->tiles[i].requested;
    }
    if (origins) memcpy(origins, t->origins, (size_t)t->sides * t->lod_count * 2 * sizeof(uint32_t));
}

/* util.rs:8-10 */
static float inverse_mix(float a, float b, float value) {
    float q = (value - a) / (b - a);
    return q < 0.0f ? 0.0f : (q > 1.0f ? 1.0f : q);
}
void orc_tile_tree_compute_blend(const orc_tile_tree* t, const double sample[3], uint32_t* lod_out, float* ratio_out) { /* :223-239 */
    dvec3 p = {sample[0], sample[1], sample[2]};
    double view_distance = length(sub(t->view_world_position, p));
    float target_lod = (float)fmin(log2(t->blend_distance / view_distance), (double)t->lod_count - 0.00001);
    uint32_t lod = !(target_lod > 0.0f) ? 0u : (uint32_t)target_lod;
    float ratio = lod == 0 ? 0.0f : inverse_mix((float)lod + t->blend_range, (float)lod, target_lod);
    *lod_out = lod;
    *ratio_out = ratio;
}

typedef struct {
    uint32_t atlas_index, atlas_lod;
    float atlas_uv[2];
} tile_lookup;

static tile_lookup lookup_tile(const orc_tile_tree* t, dvec3 world_position, uint32_t tree_lod) { /* :241-266 */
    coordinate c = coordinate_from_world_position(world_position, &t->model);
    double tile_count = (double)(1u << tree_lod);
    dvec2 tree_xy = compute_tree_xy(c, tile_count);
    orc_tree_entry entry = t->data[slot_index(t, c.side, tree_lod, (uint32_t)((uint64_t)tree_xy.x % t->tree_size),
                                              (uint32_t)((uint64_t)tree_xy.y % t->tree_size))];
    tile_lookup l = {ORC_INVALID, ORC_INVALID, {0.0f, 0.0f}};
    if (entry.atlas_lod == ORC_INVALID) return l;
    double div = (double)(1u << (tree_lod - entry.atlas_lod));
    l.atlas_index = entry.atlas_index;
    l.atlas_lod = entry.atlas_lod;
    l.atlas_uv[0] = (float)fmod(tree_xy.x / div, 1.0);
    l.atlas_uv[1] = (float)fmod(tree_xy.y / div, 1.0);
    return l;
}

void orc_tile_tree_sample_attachment(const orc_tile_tree* t, uint32_t format, uint32_t texture_size, uint32_t border_size,
                                     const void* const* layers, uint32_t atlas_size, const double* positions, uint32_t n,
                                     float* out_vec4, float* heights) { /* terrain_data/mod.rs:265-307 */
    for (uint32_t i = 0; i < n; i++) {
        dvec3 sample = {positions[3 * i], positions[3 * i + 1], positions[3 * i + 2]};
        dvec3 surface = surface_position(&t->model, sample, (double)t->approximate_height);
        double sp[3] = {surface.x, surface.y, surface.z};
        uint32_t lod;
        float blend_ratio;
        orc_tile_tree_compute_blend(t, sp, &lod, &blend_ratio);
        float value[4] = {0, 0, 0, 0};
        tile_lookup lookup = lookup_tile(t, surface, lod);
        if (lookup.atlas_index != ORC_INVALID && lookup.atlas_index < atlas_size && layers[lookup.atlas_index])
            orc_sample_tile(format, texture_size, border_size, layers[lookup.atlas_index], lookup.atlas_uv, value);
        if (blend_ratio > 0.0f) {
            float value2[4] = {0, 0, 0, 0};
            tile_lookup lookup2 = lookup_tile(t, surface, lod - 1);
            if (lookup2.atlas_index != ORC_INVALID && lookup2.atlas_index < atlas_size && layers[lookup2.atlas_index])
                orc_sample_tile(format, texture_size, border_size, layers[lookup2.atlas_index], lookup2.atlas_uv, value2);
            for (int k = 0; k < 4; k++) value[k] = value[k] + (value2[k] - value[k]) * blend_ratio; /* Vec4::lerp */
        }
        for (int k = 0; k < 4; k++) out_vec4[4 * i + k] = value[k];
        /* f32::lerp(min_height, max_height, value.x) = a + (b - a) * s */
        if (heights) heights[i] = t->model.min_height + (t->model.max_height - t->model.min_height) * value[0];
    }
}
