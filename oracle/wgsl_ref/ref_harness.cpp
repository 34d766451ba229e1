// ref_harness.cpp — runs the reference's WGSL (translated by wgsl2cpp.py into oracle/_ref/gen_*.inc) on the CPU.
// TEST INFRASTRUCTURE ONLY: built into oracle/_ref/libbt_wgslref.so, loaded by tests/, bench.py's checker legs and
// __graft_entry__.smoke() only.  Nothing under bevy_terrain_amd/ or include/ may load it.
//
// This file supplies what wgpu / Bevy supply around a compute shader and nothing more:
//   * the bindings (uniform structs, storage buffers, textures, sampler) filled the way the reference's Rust host fills
//     them (each site cites the Rust file:line it restates);
//   * textureLoad / textureSampleLevel / textureGather over R16Unorm / Rgba8Unorm data;
//   * the dispatch loop: workgroup_count x workgroup_size invocations, run sequentially in increasing
//     global_invocation_id order (atomics are therefore plain read-modify-writes: ONE of the orders a GPU may take);
//   * the copy of an atlas layer into the write section before a dispatch and back after it
//     (preprocess/mod.rs:169-210, gpu_tile_atlas.rs:285-307).
// The shader arithmetic itself is in the generated files, i.e. in the reference's own text.
//
// Texture unit definition (the one piece of arithmetic WGSL leaves to the GPU): unorm texel -> f32 is
// one correctly rounded division by 65535 / 255; R16Unorm reads as (r, 0, 0, 1); a linear sampler with
// clamp-to-edge addressing (ImageSampler::linear(), preprocessor.rs:409) computes, in f32,
// q = uv * size - 0.5, i = floor(q), f = q - i, texels at clamp(i + {0, 1}) and
// mix(mix(t00, t10, f.x), mix(t01, t11, f.x), f.y) with mix(a, b, t) = a * (1 - t) + b * t;
// textureGather returns the same four footprint texels; out-of-range textureLoad returns zero.
#include "wgsl_rt.hpp"

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../bt_oracle.h"

namespace wgsl {

struct sampler {};

struct texel_image {  // one 2D image or an array of equally sized layers, tightly packed
    const uint8_t* data = nullptr;
    uint32_t format = ORC_FORMAT_R16, width = 0, height = 0, layers = 1;
    vec4<f32> texel(uint32_t x, uint32_t y, uint32_t layer) const {
        const size_t index = (size_t(layer) * height + y) * width + x;
        if (format == ORC_FORMAT_R16) {
            uint16_t t;
            std::memcpy(&t, data + index * 2, 2);
            return vec4<f32>(f32(t) / 65535.0f, 0.0f, 0.0f, 1.0f);
        }
        const uint8_t* p = data + index * 4;
        return vec4<f32>(f32(p[0]) / 255.0f, f32(p[1]) / 255.0f, f32(p[2]) / 255.0f, f32(p[3]) / 255.0f);
    }
};
template <class T> struct texture_2d : texel_image {};
template <class T> struct texture_2d_array : texel_image {};

template <class L, class V> vec4<f32> w_textureLoad(const texture_2d_array<f32>& t, const vec2<u32>& c, L layer, V /*level*/) {
    const u32 l = w_cast<u32>(layer);
    if (c.x >= t.width || c.y >= t.height || l >= t.layers) return vec4<f32>(0.0f);
    return t.texel(c.x, c.y, l);
}

struct footprint {
    uint32_t x0, x1, y0, y1;
    f32 fx, fy;
};
inline footprint bilinear_footprint(const texel_image& t, const vec2<f32>& uv) {
    const f32 qx = uv.x * f32(t.width) - 0.5f, qy = uv.y * f32(t.height) - 0.5f;
    const f32 ix = std::floor(qx), iy = std::floor(qy);
    auto clampi = [](long long v, long long hi) { return uint32_t(v < 0 ? 0 : (v > hi ? hi : v)); };
    // uv far outside [0, 1] (or NaN) would overflow the integer conversion: clamp in f32 first, as clamp-to-edge does
    auto tol = [](f32 v) { return v != v ? 0ll : (long long)(v < -4.0e9f ? -4.0e9f : (v > 4.0e9f ? 4.0e9f : v)); };
    footprint f;
    f.fx = qx - ix;
    f.fy = qy - iy;
    f.x0 = clampi(tol(ix), t.width - 1);
    f.x1 = clampi(tol(ix) + 1, t.width - 1);
    f.y0 = clampi(tol(iy), t.height - 1);
    f.y1 = clampi(tol(iy) + 1, t.height - 1);
    return f;
}
template <class V> vec4<f32> w_textureSampleLevel(const texture_2d<f32>& t, const sampler&, const vec2<f32>& uv, V /*level*/) {
    const footprint f = bilinear_footprint(t, uv);
    const vec4<f32> t00 = t.texel(f.x0, f.y0, 0), t10 = t.texel(f.x1, f.y0, 0), t01 = t.texel(f.x0, f.y1, 0), t11 = t.texel(f.x1, f.y1, 0);
    return w_mix(w_mix(t00, t10, f.fx), w_mix(t01, t11, f.fx), f.fy);
}
// component `c` of the four footprint texels in WGSL's order (u_min v_max, u_max v_max, u_max v_min, u_min v_min)
template <class C> vec4<f32> w_textureGather(C component, const texture_2d<f32>& t, const sampler&, const vec2<f32>& uv) {
    const footprint f = bilinear_footprint(t, uv);
    const int c = int(w_cast<u32>(component));
    return vec4<f32>(t.texel(f.x0, f.y1, 0)[c], t.texel(f.x1, f.y1, 0)[c], t.texel(f.x1, f.y0, 0)[c], t.texel(f.x0, f.y0, 0)[c]);
}

#include "../_ref/gen_selftest.inc"
#include "../_ref/gen_selftest_twice.inc"
#include "../_ref/gen_split.inc"
#include "../_ref/gen_downsample.inc"
#include "../_ref/gen_stitch.inc"
#include "../_ref/gen_prepass_planar.inc"
#include "../_ref/gen_prepass_spherical.inc"

}  // namespace wgsl

using namespace wgsl;

namespace {

// AtlasBufferInfo::new — terrain_data/gpu_tile_atlas.rs:80-123
struct buffer_info {
    uint32_t pixel_size, pixels_per_entry, entries_per_side, entries_per_tile, workgroups_x, workgroups_y;
};
uint32_t align_byte_size(uint32_t value) { return value - 1 - (value - 1) % 256 + 256; }  // gpu_tile_atlas.rs:25-28
buffer_info make_buffer_info(uint32_t format, uint32_t texture_size) {
    buffer_info b;
    b.pixel_size = format == ORC_FORMAT_R16 ? 2 : 4;  // terrain_data/mod.rs:76
    b.pixels_per_entry = 4 / b.pixel_size;
    const uint32_t aligned_side_size = align_byte_size(texture_size * b.pixel_size);
    b.entries_per_side = aligned_side_size / 4;
    b.entries_per_tile = texture_size * b.entries_per_side;
    b.workgroups_x = b.entries_per_side / 8;
    b.workgroups_y = texture_size / 8;
    return b;
}

template <class P> void fill_attachment(P& p, const orc_task_desc* d, const buffer_info& b) {
    // AttachmentMeta — gpu_tile_atlas.rs:165-176
    p.attachment__preprocessing.format_id = d->format;
    p.attachment__preprocessing.lod_count = d->lod_count;
    p.attachment__preprocessing.texture_size = d->texture_size;
    p.attachment__preprocessing.border_size = d->border_size;
    p.attachment__preprocessing.center_size = d->texture_size - 2 * d->border_size;
    p.attachment__preprocessing.pixels_per_entry = b.pixels_per_entry;
    p.attachment__preprocessing.entries_per_side = b.entries_per_side;
    p.attachment__preprocessing.entries_per_tile = b.entries_per_tile;
    p.atlas__preprocessing.data = static_cast<const uint8_t*>(d->atlas);
    p.atlas__preprocessing.format = d->format;
    p.atlas__preprocessing.width = p.atlas__preprocessing.height = d->texture_size;
    p.atlas__preprocessing.layers = d->atlas_size;
}
template <class A> A make_atlas_tile(const orc_atlas_tile& t) {  // AtlasTile — tile_atlas.rs:30-35
    A a;
    a.coordinate.side = t.coordinate.side;
    a.coordinate.lod = t.coordinate.lod;
    a.coordinate.x = t.coordinate.x;
    a.coordinate.y = t.coordinate.y;
    a.atlas_index = t.atlas_index;
    return a;
}

template <class P, class F> void dispatch(P& p, const unsigned (&wg)[3], uint32_t gx, uint32_t gy, uint32_t gz, F entry) {
    for (uint32_t wz = 0; wz < gz; wz++)
        for (uint32_t wy = 0; wy < gy; wy++)
            for (uint32_t wx = 0; wx < gx; wx++)
                for (uint32_t lz = 0; lz < wg[2]; lz++)
                    for (uint32_t ly = 0; ly < wg[1]; ly++)
                        for (uint32_t lx = 0; lx < wg[0]; lx++) (p.*entry)(vec3<u32>(wx * wg[0] + lx, wy * wg[1] + ly, wz * wg[2] + lz));
}

// one preprocessing task = copy_tiles_to_write_section (slot 0) + one dispatch + copy_tiles_from_write_section
template <class P, class F> void run_preprocess(P& p, const orc_task_desc* d, const unsigned (&wg)[3], F entry, void* out_tile) {
    const buffer_info b = make_buffer_info(d->format, d->texture_size);
    fill_attachment(p, d, b);
    const uint32_t T = d->texture_size;
    const size_t row_bytes = size_t(T) * b.pixel_size;
    std::vector<u32> section(b.entries_per_tile, 0u);
    if (d->tile.atlas_index < d->atlas_size) {  // the layer travels into the slot with padded rows
        const uint8_t* layer = static_cast<const uint8_t*>(d->atlas) + size_t(d->tile.atlas_index) * T * row_bytes;
        for (uint32_t y = 0; y < T; y++) std::memcpy(reinterpret_cast<uint8_t*>(section.data() + size_t(y) * b.entries_per_side), layer + y * row_bytes, row_bytes);
    }
    p.atlas_write_section__preprocessing.data = section.data();
    p.atlas_write_section__preprocessing.size = section.size();
    dispatch(p, wg, b.workgroups_x, b.workgroups_y, 1, entry);
    uint8_t* out = static_cast<uint8_t*>(out_tile);
    for (uint32_t y = 0; y < T; y++) std::memcpy(out + y * row_bytes, reinterpret_cast<const uint8_t*>(section.data() + size_t(y) * b.entries_per_side), row_bytes);
}

}  // namespace

extern "C" {

// signature = orc_task_backend (bt_oracle.h): the oracle's queue driver hands every Split / Downsample / Stitch task
// here instead of to its own restated kernels
void wref_run_task(void* /*user*/, const orc_task_desc* d, void* out_tile) {
    if (d->type == ORC_TASK_SPLIT) {
        Split p;
        // SplitData — preprocess/preprocess/gpu_preprocessor.rs:31-36,155-162 (tile_index = the write-section slot)
        p.split_data__split.tile = make_atlas_tile<Split::AtlasTile__preprocessing>(d->tile);
        p.split_data__split.top_left = vec2<f32>(d->top_left[0], d->top_left[1]);
        p.split_data__split.bottom_right = vec2<f32>(d->bottom_right[0], d->bottom_right[1]);
        p.split_data__split.tile_index = 0;
        p.source_tile__split.data = static_cast<const uint8_t*>(d->src);
        p.source_tile__split.format = d->format;
        p.source_tile__split.width = d->src_w;
        p.source_tile__split.height = d->src_h;
        run_preprocess(p, d, Split::split__split_workgroup_size, &Split::split__split, out_tile);
    } else if (d->type == ORC_TASK_DOWNSAMPLE) {
        Downsample p;
        // DownsampleData — gpu_preprocessor.rs:46-50,196-203
        p.downsample_data__downsample.tile = make_atlas_tile<Downsample::AtlasTile__preprocessing>(d->tile);
        for (int i = 0; i < 4; i++) p.downsample_data__downsample.child_tiles[i] = make_atlas_tile<Downsample::AtlasTile__preprocessing>(d->rel[i]);
        p.downsample_data__downsample.tile_index = 0;
        run_preprocess(p, d, Downsample::downsample__downsample_workgroup_size, &Downsample::downsample__downsample, out_tile);
    } else if (d->type == ORC_TASK_STITCH) {
        Stitch p;
        // StitchData — gpu_preprocessor.rs:39-43,178-185
        p.stitch_data__stitch.tile = make_atlas_tile<Stitch::AtlasTile__preprocessing>(d->tile);
        for (int i = 0; i < 8; i++) p.stitch_data__stitch.neighbour_tiles[i] = make_atlas_tile<Stitch::AtlasTile__preprocessing>(d->rel[i]);
        p.stitch_data__stitch.tile_index = 0;
        run_preprocess(p, d, Stitch::stitch__stitch_workgroup_size, &Stitch::stitch__stitch, out_tile);
    }
}

}  // extern "C"

namespace {

template <class P> void fill_view(P& p, const orc_view* v, std::vector<typename std::remove_pointer_t<decltype(P::mesh__bindings.data)>>& mesh) {
    // TerrainViewConfigUniform — terrain_view_bind_group.rs:81-116 (fields the prepass reads)
    p.view_config__bindings.tile_count = v->tile_count;
    p.view_config__bindings.refinement_count = v->refinement_count;
    p.view_config__bindings.vertices_per_tile = v->vertices_per_tile;
    p.view_config__bindings.subdivision_distance = v->subdivision_distance;
    // TerrainModelApproximation — terrain_model.rs:228-259
    p.terrain_model_approximation__bindings.origin_lod = v->origin_lod;
    p.terrain_model_approximation__bindings.approximate_height = v->approximate_height;
    for (int s = 0; s < 6; s++) {
        p.terrain_model_approximation__bindings.sides[s].view_xy = vec2<i32>(v->sides[s].view_xy[0], v->sides[s].view_xy[1]);
        p.terrain_model_approximation__bindings.sides[s].view_uv = vec2<f32>(v->sides[s].view_uv[0], v->sides[s].view_uv[1]);
    }
    // CullingUniform.world_position — culling_bind_group.rs:50
    p.culling_view__bindings.world_position = vec3<f32>(v->world_position[0], v->world_position[1], v->world_position[2]);
    // MeshUniform (bevy_pbr 0.14 MeshUniform::new): world_from_local = Affine3::to_transpose() — three vec4 ROWS of the
    // 3x4 affine; local_from_world_transpose packed as two vec4 + one f32, column-major
    mesh.resize(1);
    const float* a = v->world_from_local;  // columns 0..2, then the translation
    for (int r = 0; r < 3; r++) mesh[0].world_from_local[r] = vec4<f32>(a[r], a[3 + r], a[6 + r], a[9 + r]);
    const float* n = v->local_from_world_transpose;
    mesh[0].local_from_world_transpose_a[0] = vec4<f32>(n[0], n[1], n[2], n[3]);
    mesh[0].local_from_world_transpose_a[1] = vec4<f32>(n[4], n[5], n[6], n[7]);
    mesh[0].local_from_world_transpose_b = n[8];
    p.mesh__bindings.data = mesh.data();
    p.mesh__bindings.size = mesh.size();
}

// TilingPrepassNode::run — render/tiling_prepass.rs:244-263
template <class P> long run_prepass(const orc_view* v, orc_coord* final_tiles, uint32_t cap, uint32_t indirect[4], uint32_t* passes_tile_counts) {
    P p;
    using Tile = std::remove_pointer_t<decltype(p.temporary_tiles__bindings.data)>;
    std::vector<std::remove_pointer_t<decltype(p.mesh__bindings.data)>> mesh;
    fill_view(p, v, mesh);
    std::vector<Tile> temporary(v->tile_count), final_list(v->tile_count);
    p.temporary_tiles__bindings.data = temporary.data();
    p.temporary_tiles__bindings.size = temporary.size();
    p.final_tiles__bindings.data = final_list.data();
    p.final_tiles__bindings.size = final_list.size();

    auto refine = [&](uint32_t pass) {
        if (passes_tile_counts) passes_tile_counts[pass] = p.parameters__bindings.tile_count;
        const auto wc = p.indirect_buffer__bindings.workgroup_count;  // dispatch_workgroups_indirect
        dispatch(p, P::refine_tiles__refine_tiles_workgroup_size, wc.x, wc.y, wc.z, &P::refine_tiles__refine_tiles);
    };
    p.prepare_root__prepare_prepass();
    for (uint32_t i = 0; i < v->refinement_count; i++) {
        refine(i);
        p.prepare_next__prepare_prepass();
    }
    refine(v->refinement_count);
    p.prepare_render__prepare_prepass();

    const uint32_t count = uint32_t(p.parameters__bindings.final_index.v);
    if (indirect) {
        indirect[0] = p.indirect_buffer__bindings.workgroup_count.x;
        indirect[1] = p.indirect_buffer__bindings.workgroup_count.y;
        indirect[2] = p.indirect_buffer__bindings.workgroup_count.z;
        indirect[3] = 0;
    }
    if (p.temporary_tiles__bindings.overflow || p.final_tiles__bindings.overflow || count > cap) return -1;
    for (uint32_t i = 0; i < count; i++) final_tiles[i] = orc_coord{final_list[i].side, final_list[i].lod, final_list[i].xy.x, final_list[i].xy.y};
    return long(count);
}

template <class P> int should_be_divided(const orc_view* v, orc_coord tile, float* view_distance) {
    P p;
    std::vector<std::remove_pointer_t<decltype(p.mesh__bindings.data)>> mesh;
    fill_view(p, v, mesh);
    typename P::TileCoordinate__types t{tile.side, tile.lod, vec2<u32>(tile.x, tile.y)};
    if (view_distance) {
        const auto c = p.compute_subdivision_coordinate__functions(typename P::Coordinate__types{t.side, t.lod, t.xy, vec2<f32>(0.0f)});
        *view_distance = p.approximate_view_distance__functions(c, p.culling_view__bindings.world_position);
    }
    return p.should_be_divided__refine_tiles(t) ? 1 : 0;
}

}  // namespace

extern "C" {

// same contract as orc_refine (bt_oracle.h)
long wref_refine(const orc_view* v, orc_coord* final_tiles, uint32_t cap, uint32_t indirect[4], uint32_t* passes_tile_counts) {
    return v->spherical ? run_prepass<PrepassSpherical>(v, final_tiles, cap, indirect, passes_tile_counts)
                        : run_prepass<PrepassPlanar>(v, final_tiles, cap, indirect, passes_tile_counts);
}
int wref_should_be_divided(const orc_view* v, orc_coord tile, float* view_distance) {
    return v->spherical ? should_be_divided<PrepassSpherical>(v, tile, view_distance) : should_be_divided<PrepassPlanar>(v, tile, view_distance);
}
// oracle/wgsl_ref/selftest/selftest.wgsl: the translator's known-answer shader (out_f: 32 floats, out_u: 32 words)
void wref_selftest(int twice, float* out_f, uint32_t* out_u) {
    auto run = [&](auto& p) {
        p.out_f__lib.data = out_f;
        p.out_f__lib.size = 32;
        p.out_u__lib.data = out_u;
        p.out_u__lib.size = 32;
        dispatch(p, std::remove_reference_t<decltype(p)>::selftest__selftest_workgroup_size, 1, 1, 1, &std::remove_reference_t<decltype(p)>::selftest__selftest);
    };
    if (twice) { SelfTestTwice p; run(p); } else { SelfTest p; run(p); }
}
// the shader sources this library was generated from (for test reports)
const char* wref_sources(void) { return WREF_SOURCES; }

}  // extern "C"
