#define_import_path selftest::lib

// Self-test module for the translator (OURS, not the reference's): naga_oil features in miniature.
struct Pair {
    a: u32,
    xy: vec2<u32>,
}

const K = 0.1 + 0.2;          // abstract: evaluated in f64, concretised to f32 at the declaration (naga 0.20 rule)
const N: u32 = 3u;

@group(0) @binding(0)
var<storage, read_write> out_f: array<f32>;
@group(0) @binding(1)
var<storage, read_write> out_u: array<u32>;

virtual fn flavour(x: u32) -> u32 { return x + 1u; }

fn uses_flavour(x: u32) -> u32 { return flavour(x) * 10u; }

fn bump(p: ptr<function, Pair>, by: u32) {
    (*p).a = (*p).a + by;
    (*p).xy = (*p).xy * 2u + vec2<u32>(by);
}

#ifdef TWICE
fn scale() -> f32 { return 2.0; }
#else
fn scale() -> f32 { return 1.0; }
#endif
