// f64 terrain maths shared by the host (view-state derivation, tile streaming) and the device (TileTree kernels):
// TerrainModel (math/terrain_model.rs:41-220), Coordinate (math/coordinate.rs:57-151), the ellipsoid projection
// (math/ellipsoid.rs) and the per-node helpers of TileTree (terrain_data/tile_tree.rs:175-266).
//
// Every function is plain IEEE binary64 +, -, *, /, sqrt, floor / trunc / round (compiled with -ffp-contract=off), so
// the host, the device and the oracle's C restatement produce the same bits.  Where the reference defers to a library
// the definition used here is:
//   * DMat4::from_scale_rotation_translation(scale, IDENTITY, t) and its `.inverse()` (glam's general cofactor inverse):
//     world = scale * local + t, local = (world - t) / scale, component-wise;
//   * `.powf(0.5)` (libm pow): sqrt;
//   * DVec3::normalize(): v * (1.0 / length) (glam), length = sqrt((x*x + y*y) + z*z).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>

#include "bevy_terrain_amd.h"

#define BT_HD __host__ __device__ __forceinline__

namespace bt {
namespace model {

constexpr double kCSqr = 0.87 * 0.87;  // math/mod.rs:13

struct V3 {
    double x, y, z;
};
struct V2 {
    double x, y;
};

BT_HD double dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
BT_HD double length3(V3 a) { return sqrt(dot3(a, a)); }
BT_HD V3 normalize3(V3 a) {
    const double r = 1.0 / length3(a);
    return {a.x * r, a.y * r, a.z * r};
}
BT_HD double distance3(V3 a, V3 b) { return length3({a.x - b.x, a.y - b.y, a.z - b.z}); }

// TerrainModel with the reference constructors' identity rotation
struct Model {
    uint32_t kind;
    V3 position, scale;
    double a, b;
    float min_height, max_height;
};

BT_HD Model make_model(const bt_terrain_model& m) {
    Model o;
    o.kind = m.kind;
    o.position = {m.position[0], m.position[1], m.position[2]};
    // planar / sphere: DVec3::splat; ellipsoid: (major, minor, major) (terrain_model.rs:96-138)
    o.scale = m.kind == BT_MODEL_ELLIPSOIDAL ? V3{m.a, m.b, m.a} : V3{m.a, m.a, m.a};
    o.a = m.a;
    o.b = m.b;
    o.min_height = m.min_height;
    o.max_height = m.max_height;
    return o;
}
BT_HD bool is_spherical(const Model& m) { return m.kind != BT_MODEL_PLANAR; }
BT_HD uint32_t side_count(const Model& m) { return is_spherical(m) ? 6u : 1u; }
// terrain_model.rs:183-193
BT_HD double model_scale(const Model& m) {
    return m.kind == BT_MODEL_PLANAR ? m.a / 2.0 : (m.kind == BT_MODEL_SPHERICAL ? m.a : (m.a + m.b) / 2.0);
}

BT_HD V3 transform_point(const Model& m, V3 p) { return {m.scale.x * p.x + m.position.x, m.scale.y * p.y + m.position.y, m.scale.z * p.z + m.position.z}; }
BT_HD V3 transform_vector(const Model& m, V3 v) { return {m.scale.x * v.x, m.scale.y * v.y, m.scale.z * v.z}; }
BT_HD V3 inverse_transform_point(const Model& m, V3 p) {
    return {(p.x - m.position.x) / m.scale.x, (p.y - m.position.y) / m.scale.y, (p.z - m.position.z) / m.scale.z};
}

// terrain_model.rs:140-152
BT_HD V3 position_local_to_world(const Model& m, V3 local, double height) {
    const V3 w = transform_point(m, local);
    const V3 n = normalize3(transform_vector(m, is_spherical(m) ? local : V3{0.0, 1.0, 0.0}));
    return {w.x + height * n.x, w.y + height * n.y, w.z + height * n.z};
}

// ---- math/ellipsoid.rs (bisection root finders; MAX_ITERATIONS = 1074) ---------------------------------------
BT_HD double get_root_2d(V2 r, V2 z, double g) {
    const V2 n = {r.x * z.x, r.y * z.y};
    double s0 = z.y - 1.0;
    double s1 = g < 0.0 ? 0.0 : sqrt(n.x * n.x + n.y * n.y) - 1.0;
    double s = 0.0;
    for (int i = 0; i < 1074; i++) {
        s = (s0 + s1) / 2.0;
        if (s == s0 || s == s1) break;
        const V2 ratio = {n.x / (s + r.x), n.y / (s + r.y)};
        const double gg = (ratio.x * ratio.x + ratio.y * ratio.y) - 1.0;
        if (gg < 0.0) s1 = s;
        else if (gg > 0.0) s0 = s;
        else break;
    }
    return s;
}
BT_HD double get_root_3d(V3 r, V3 z, double g) {
    const V3 n = {r.x * z.x, r.y * z.y, r.z * z.z};
    double s0 = z.z - 1.0;
    double s1 = g < 0.0 ? 0.0 : length3(n) - 1.0;
    double s = 0.0;
    for (int i = 0; i < 1074; i++) {
        s = (s0 + s1) / 2.0;
        if (s == s0 || s == s1) break;
        const V3 ratio = {n.x / (s + r.x), n.y / (s + r.y), n.z / (s + r.z)};
        const double gg = dot3(ratio, ratio) - 1.0;
        if (gg < 0.0) s1 = s;
        else if (gg > 0.0) s0 = s;
        else break;
    }
    return s;
}
BT_HD V2 project_point_ellipse(V2 e, V2 y) {
    if (y.y > 0.0) {
        if (y.x > 0.0) {
            const V2 z = {y.x / e.x, y.y / e.y};
            const double g = (z.x * z.x + z.y * z.y) - 1.0;
            if (g != 0.0) {
                const V2 r = {(e.x * e.x) / (e.y * e.y), 1.0};
                const double t = get_root_2d(r, z, g);
                return {r.x * y.x / (t + r.x), r.y * y.y / (t + r.y)};
            }
            return y;
        }
        return {0.0, e.y};
    }
    const double numer0 = e.x * y.x, denom0 = e.x * e.x - e.y * e.y;
    if (numer0 < denom0) {
        const double xde0 = numer0 / denom0;
        return {e.x * xde0, e.y * sqrt(1.0 - xde0 * xde0)};
    }
    return {e.x, 0.0};
}
BT_HD double signum(double v) { return v != v ? v : (__builtin_signbit(v) ? -1.0 : 1.0); }  // f64::signum: -1 for -0.0 too
// ellipsoid.rs:12-62.  e = semi-axes, y = query point; both in the frame of the call site (terrain_model.rs:163-171
// passes e = (major, major, minor)); the function swizzles y to xzy internally and back at the end.
BT_HD V3 project_point_ellipsoid(V3 e, V3 yin) {
    const V3 sign = {signum(yin.x), signum(yin.y), signum(yin.z)};
    const V3 y = {fabs(yin.x), fabs(yin.z), fabs(yin.y)};  // y.xzy().abs()
    V3 x;
    if (y.z > 0.0) {
        if (y.y > 0.0) {
            if (y.x > 0.0) {
                const V3 z = {y.x / e.x, y.y / e.y, y.z / e.z};
                const double g = dot3(z, z) - 1.0;
                if (g != 0.0) {
                    const V3 r = {(e.x * e.x) / (e.z * e.z), (e.y * e.y) / (e.z * e.z), 1.0};
                    const double t = get_root_3d(r, z, g);
                    x = {r.x * y.x / (t + r.x), r.y * y.y / (t + r.y), r.z * y.z / (t + r.z)};
                } else {
                    x = y;
                }
            } else {
                const V2 p = project_point_ellipse({e.y, e.z}, {y.y, y.z});
                x = {0.0, p.x, p.y};  // (p, 0).zxy()
            }
        } else {
            if (y.x > 0.0) {
                const V2 p = project_point_ellipse({e.x, e.z}, {y.x, y.z});
                x = {p.x, 0.0, p.y};  // (p, 0).xzy()
            } else {
                x = {0.0, 0.0, e.z};
            }
        }
    } else {
        const double denom0 = e.x * e.x - e.z * e.z, denom1 = e.y * e.y - e.z * e.z;
        const double numer0 = e.x * y.x, numer1 = e.y * y.y;
        bool found = false;
        if (numer0 < denom0 && numer1 < denom1) {
            const double xde0 = numer0 / denom0, xde1 = numer1 / denom1;
            const double discr = (1.0 - xde0 * xde0) - xde1 * xde1;
            if (discr > 0.0) {
                x = {e.x * xde0, e.y * xde1, e.z * sqrt(discr)};
                found = true;
            }
        }
        if (!found) {
            const V2 p = project_point_ellipse({e.x, e.y}, {y.x, y.y});
            x = {p.x, p.y, 0.0};
        }
    }
    return {sign.x * x.x, sign.y * x.z, sign.z * x.y};  // sign * x.xzy()
}

// terrain_model.rs:154-176
BT_HD V3 position_world_to_local(const Model& m, V3 world) {
    if (m.kind == BT_MODEL_PLANAR) {
        const V3 p = inverse_transform_point(m, world);
        return {1.0 * p.x, 0.0 * p.y, 1.0 * p.z};
    }
    if (m.kind == BT_MODEL_SPHERICAL) return normalize3(inverse_transform_point(m, world));
    // ellipsoid_from_world = inverse of (identity rotation, translation): world - position
    const V3 ellipsoid_position = {world.x - m.position.x, world.y - m.position.y, world.z - m.position.z};
    const V3 surface = project_point_ellipsoid({m.a, m.a, m.b}, ellipsoid_position);
    return normalize3(inverse_transform_point(m, surface));
}
// terrain_model.rs:178-180
BT_HD V3 surface_position(const Model& m, V3 world, double height) { return position_local_to_world(m, position_world_to_local(m, world), height); }

// ---- Coordinate (math/coordinate.rs:57-151) ----------------------------------------------------------------
struct Coordinate {
    uint32_t side;
    V2 uv;
};

BT_HD double clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }

// coordinate.rs:69-113
BT_HD Coordinate coordinate_from_world_position(V3 world, const Model& m) {
    const V3 p = position_world_to_local(m, world);
    if (!is_spherical(m)) return {0u, {clamp01(p.x + 0.5), clamp01(p.z + 0.5)}};
    const double ax = fabs(p.x), ay = fabs(p.y), az = fabs(p.z);
    uint32_t side;
    V2 uv;
    if (ax > ay && ax > az) {
        if (p.x < 0.0) { side = 0; uv = {-p.z / p.x, p.y / p.x}; } else { side = 3; uv = {-p.y / p.x, p.z / p.x}; }
    } else if (az > ay) {
        if (p.z > 0.0) { side = 1; uv = {p.x / p.z, -p.y / p.z}; } else { side = 4; uv = {p.y / p.z, -p.x / p.z}; }
    } else {
        if (p.y > 0.0) { side = 2; uv = {p.x / p.y, p.z / p.y}; } else { side = 5; uv = {-p.z / p.y, -p.x / p.y}; }
    }
    const V2 w = {uv.x * sqrt((1.0 + kCSqr) / (1.0 + kCSqr * uv.x * uv.x)), uv.y * sqrt((1.0 + kCSqr) / (1.0 + kCSqr * uv.y * uv.y))};
    return {side, {0.5 * w.x + 0.5, 0.5 * w.y + 0.5}};
}

// coordinate.rs:115-135
BT_HD V3 coordinate_world_position(Coordinate c, const Model& m, float height) {
    V3 local;
    if (is_spherical(m)) {
        const V2 w = {(c.uv.x - 0.5) / 0.5, (c.uv.y - 0.5) / 0.5};
        const V2 uv = {w.x / sqrt(1.0 + kCSqr - kCSqr * w.x * w.x), w.y / sqrt(1.0 + kCSqr - kCSqr * w.y * w.y)};
        switch (c.side) {
            case 0: local = {-1.0, -uv.y, uv.x}; break;
            case 1: local = {uv.x, -uv.y, 1.0}; break;
            case 2: local = {uv.x, 1.0, uv.y}; break;
            case 3: local = {1.0, -uv.x, uv.y}; break;
            case 4: local = {uv.y, -uv.x, -1.0}; break;
            default: local = {uv.y, -1.0, uv.x}; break;
        }
        local = normalize3(local);
    } else {
        local = {c.uv.x - 0.5, 0.0, c.uv.y - 0.5};
    }
    return position_local_to_world(m, local, double(height));
}

// coordinate.rs:27-53, 137-151 (SideInfo: 0 = Fixed0, 1 = Fixed1, 2 = PositiveS, 3 = PositiveT)
BT_HD Coordinate coordinate_project_to_side(Coordinate c, uint32_t side, const Model& m) {
    if (!is_spherical(m)) return c;
    const uint8_t even[6][2] = {{2, 3}, {0, 3}, {0, 2}, {3, 2}, {3, 0}, {2, 0}};
    const uint8_t odd[6][2] = {{2, 3}, {2, 1}, {3, 1}, {3, 2}, {1, 2}, {1, 3}};
    const uint32_t index = (6u + side - c.side) % 6u;
    const uint8_t i0 = (c.side % 2u == 0 ? even : odd)[index][0], i1 = (c.side % 2u == 0 ? even : odd)[index][1];
    auto pick = [&](uint8_t i) -> double { return i == 0 ? 0.0 : (i == 1 ? 1.0 : (i == 2 ? c.uv.x : c.uv.y)); };
    return {side, {pick(i0), pick(i1)}};
}

// ---- TileTree helpers (terrain_data/tile_tree.rs) ---------------------------------------------------------
// `as u32` / as_uvec2: saturating, NaN -> 0
BT_HD uint32_t saturating_u32(double v) { return !(v > 0.0) ? 0u : (v >= 4294967295.0 ? 0xFFFFFFFFu : uint32_t(v)); }
BT_HD int32_t saturating_i32(double v) { return v != v ? 0 : (v <= -2147483648.0 ? INT32_MIN : (v >= 2147483647.0 ? INT32_MAX : int32_t(v))); }

// :175-178
BT_HD V2 compute_tree_xy(Coordinate c, double tile_count) {
    const double cap = tile_count - 0.000001;
    const V2 s = {c.uv.x * tile_count, c.uv.y * tile_count};
    return {s.x < cap ? s.x : cap, s.y < cap ? s.y : cap};  // DVec2::min
}
// :180-192 (f64::round: half away from zero; DVec2::clamp = max(min).min(max); as_uvec2 saturates)
BT_HD void compute_origin(Coordinate c, uint32_t lod, uint32_t tree_size, uint32_t out[2]) {
    const double tile_count = double(1u << lod);
    const V2 t = compute_tree_xy(c, tile_count);
    const double hi = tile_count - double(tree_size);
    double v[2] = {round(t.x - 0.5 * double(tree_size)), round(t.y - 0.5 * double(tree_size))};
    for (int k = 0; k < 2; k++) {
        v[k] = v[k] > 0.0 ? v[k] : 0.0;
        v[k] = v[k] < hi ? v[k] : hi;
        out[k] = saturating_u32(v[k]);
    }
}
// :194-221
BT_HD double compute_tile_distance(bt_tile_coordinate tile, Coordinate view_coordinate, const Model& m, float approximate_height, V3 view_world_position) {
    const double tile_count = double(1u << tile.lod);
    const int32_t tx = int32_t(tile.x), ty = int32_t(tile.y);
    const V2 v = compute_tree_xy(view_coordinate, tile_count);
    const int32_t ox = saturating_i32(v.x) - tx, oy = saturating_i32(v.y) - ty;
    V2 offset = {v.x - trunc(v.x), v.y - trunc(v.y)};  // `% 1.0` for non-negative values
    if (ox < 0) offset.x = 0.0; else if (ox > 0) offset.x = 1.0;
    if (oy < 0) offset.y = 0.0; else if (oy > 0) offset.y = 1.0;
    const Coordinate c = {tile.side, {(double(tx) + offset.x) / tile_count, (double(ty) + offset.y) / tile_count}};
    return distance3(coordinate_world_position(c, m, approximate_height), view_world_position);
}

}  // namespace model
}  // namespace bt
