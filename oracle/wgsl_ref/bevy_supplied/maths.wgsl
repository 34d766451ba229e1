#define_import_path bevy_render::maths

// BEVY-SUPPLIED, NOT PART OF THE REFERENCE REPOSITORY.  functions.wgsl:5 imports these two helpers from Bevy's own
// shader library (bevy_render/src/maths.wgsl of bevy 0.14.0, the version Cargo.toml:18 pins); Bevy is not vendored
// under /root/reference, so their published text is restated here for the translator.  TEST INFRASTRUCTURE ONLY.

fn affine3_to_square(affine: mat3x4<f32>) -> mat4x4<f32> {
    return transpose(mat4x4<f32>(
        affine[0],
        affine[1],
        affine[2],
        vec4<f32>(0.0, 0.0, 0.0, 1.0),
    ));
}

fn mat2x4_f32_to_mat3x3_unpack(
    a: mat2x4<f32>,
    b: f32,
) -> mat3x3<f32> {
    return mat3x3<f32>(
        a[0].xyz,
        vec3<f32>(a[0].w, a[1].xy),
        vec3<f32>(a[1].zw, b),
    );
}
