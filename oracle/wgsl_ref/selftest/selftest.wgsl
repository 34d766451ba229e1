#import selftest::lib::{Pair, K, N, out_f, out_u, flavour, uses_flavour, bump, scale}

override fn flavour(x: u32) -> u32 { return x + 7u; }

fn sum3(v: vec3<f32>) -> f32 { return v.x + v.y + v.z; }

@compute @workgroup_size(2, 1, 1)
fn selftest(@builtin(global_invocation_id) id: vec3<u32>) {
    if (id.x != 1u) { return; }   // two invocations are dispatched; only the second one writes

    // ---- floats
    out_f[0] = K;                                   // f32(0.1 + 0.2 in f64)
    let a = 0.1;                                    // let: concretised to f32
    let b = 0.2;
    out_f[1] = a + b;                               // f32 + f32
    out_f[2] = 1.0 + K;                             // K is f32 already: f32 add
    out_f[3] = 7.5 % 2.0;                           // abstract: 1.5
    let big = 7.5f;
    out_f[4] = big % 2.0;                           // f32: x - y * trunc(x / y)
    out_f[5] = -7.5f % 2.0;                         // -1.5 (sign of the dividend)
    out_f[6] = mix(2.0f, 4.0f, 0.25f);              // 2 * 0.75 + 4 * 0.25
    out_f[7] = clamp(5.0f, 0.0, 1.0) + step(0.5, 0.5f) + step(0.6, 0.5f);   // 1 + 1 + 0
    out_f[8] = pow(2.0, f32(-3));                   // 0.125
    out_f[9] = sum3(normalize(vec3<f32>(3.0, 0.0, 4.0)));   // 0.6 + 0 + 0.8 in f32
    out_f[10] = distance(vec2<f32>(1.0, 1.0), vec2<f32>(4.0, 5.0));   // 5
    let v4 = vec4<f32>(vec2<f32>(1.0, 2.0), 3.0, 4.0);
    out_f[11] = v4.w + v4.zy.x * 10.0 + v4.xyz.y * 100.0;   // 4 + 30 + 200
    out_f[12] = select(1.0f, 2.0f, v4.x < v4.y);    // true -> second
    let m = mat3x3<f32>(vec3<f32>(1.0, 2.0, 3.0), vec3<f32>(4.0, 5.0, 6.0), vec3<f32>(7.0, 8.0, 9.0));
    let mv = m * vec3<f32>(1.0, 10.0, 100.0);       // columns: (1,2,3) + 10 (4,5,6) + 100 (7,8,9)
    out_f[13] = mv.x;
    out_f[14] = transpose(m)[0].y;                  // m[1].x = 4
    out_f[15] = scale();
    out_f[16] = f32(1u << 5u) / 3.0;                // one f32 division
    out_f[17] = (vec2<f32>(vec2<u32>(3u, 5u)) / 2.0).y;
    var acc = vec2<f32>(0.0);
    for (var i = 0u; i < N; i += 1u) { acc += vec2<f32>(f32(i), 1.0); }
    out_f[18] = acc.x * 10.0 + acc.y;               // (0+1+2) * 10 + 3
    out_f[19] = sqrt(2.0f);

    // ---- integers
    let zero = 0u;
    out_u[0] = 17u / zero;                          // x / 0 = x
    out_u[1] = 17u % zero;                          // x % 0 = 0
    out_u[2] = 1u << 33u;                           // shift amount mod 32 -> 2
    let neg = -5;                                   // i32
    out_u[3] = u32(neg);                            // bit pattern
    out_u[4] = u32(-3.7f) + u32(3.99f) * 10u + u32(1.0e20f) / 0x10000000u;   // 0 + 30 + 15
    out_u[5] = u32(neg >> 1u);                      // arithmetic shift: -3
    out_u[6] = pack2x16unorm(vec2<f32>(0.5, 2.0));  // floor(0.5 + 32767.5) = 32768 | 65535 << 16
    out_u[7] = pack4x8unorm(vec4<f32>(0.0, 1.0, 0.499, -1.0));
    out_u[8] = uses_flavour(1u);                    // override: (1 + 7) * 10
    var p = Pair(1u, vec2<u32>(2u, 3u));
    bump(&p, 5u);
    out_u[9] = p.a * 10000u + p.xy.x * 100u + p.xy.y;   // 6, 9, 11
    var list = array(vec2(1u, 2u), vec2(3u, 4u), vec2(5u, 6u));
    var total = 0u;
    for (var i = 0u; i < 3u; i = i + 1u) { total += list[i].x * list[i].y; }
    out_u[10] = total;                              // 2 + 12 + 30
    var sw = 0u;
    switch (total) {
        case 44u: { sw = 1u; }
        case 1u, 2u: { sw = 2u; }
        case default: { sw = 3u; }
    }
    out_u[11] = sw;
    out_u[12] = select(10u, 20u, all(vec2<u32>(1u, 2u) != vec2<u32>(0u))) + select(1u, 2u, any(vec2<f32>(0.0) != vec2<f32>(0.0)));
    let shadow = 4u;
    { let shadow = shadow + 1u; out_u[13] = shadow; }   // initialiser sees the outer binding
    out_u[14] = shadow;
    out_u[15] = u32(i32(7u) * clamp(-4, 0, 1) - i32(3u) * -1);   // 0 + 3
    out_u[16] = (6u + 1u - 3u) % 6u + (7u & 3u) + (8u >> 1u & 1u) * 100u;   // 4 + 3 + 0
    out_u[17] = id.x;
}
