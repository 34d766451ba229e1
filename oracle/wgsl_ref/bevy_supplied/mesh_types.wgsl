#define_import_path bevy_pbr::mesh_types

// BEVY-SUPPLIED, NOT PART OF THE REFERENCE REPOSITORY.  bindings.wgsl:4 imports the `Mesh` storage-buffer element from
// bevy_pbr (bevy_pbr/src/render/mesh_types.wgsl of bevy 0.14.0); restated here for the translator.  The prepass reads
// world_from_local and local_from_world_transpose_{a,b} of mesh[0] only.  TEST INFRASTRUCTURE ONLY.

struct Mesh {
    // Affine 4x3 matrices transposed to 3x4
    world_from_local: mat3x4<f32>,
    previous_world_from_local: mat3x4<f32>,
    // 3x3 matrix packed in mat2x4 and f32 as:
    //   [0].xyz, [1].x,
    //   [1].yz, [2].xy
    //   [2].z
    local_from_world_transpose_a: mat2x4<f32>,
    local_from_world_transpose_b: f32,
    flags: u32,
    lightmap_uv_rect: vec2<u32>,
}
