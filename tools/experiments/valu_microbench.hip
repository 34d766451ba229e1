// Micro-benchmark: VALU issue cost on gfx950 (cycles per wave64 instruction per SIMD) for plain and packed f32,
// independent vs dependent chains, at 1..4 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3 valu_microbench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    f2 p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
    const f2 a2 = {a, a}, b2 = {b, b};
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) {  // 8 independent scalar fma chains
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (MODE == 1) {  // one dependent scalar chain (8 per iteration)
#pragma unroll
            for (int j = 0; j < 8; j++) x0 = __builtin_fmaf(x0, a, b);
        } else if (MODE == 2) {  // 4 independent packed chains, 8 pk instr per iteration
            p0 = __builtin_elementwise_fma(p0, a2, b2); p1 = __builtin_elementwise_fma(p1, a2, b2);
            p2 = __builtin_elementwise_fma(p2, a2, b2); p3 = __builtin_elementwise_fma(p3, a2, b2);
            p0 = __builtin_elementwise_fma(p0, a2, b2); p1 = __builtin_elementwise_fma(p1, a2, b2);
            p2 = __builtin_elementwise_fma(p2, a2, b2); p3 = __builtin_elementwise_fma(p3, a2, b2);
        } else {  // one dependent packed chain
#pragma unroll
            for (int j = 0; j < 8; j++) p0 = __builtin_elementwise_fma(p0, a2, b2);
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}
template <int MODE>
void run(const char* name, float* out, int blocks_per_cu) {
    const int iters = 20000, cus = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<cus * blocks_per_cu, 256>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<cus * blocks_per_cu, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = double(iters) * 8 * blocks_per_cu;  // one wave per SIMD per block
    printf("%-28s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (x2.4 GHz = %.2f cycles)\n", name, blocks_per_cu, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}
int main() {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 4, 6, 8}) {
        run<0>("scalar fma, independent x8", out, w);
        run<1>("scalar fma, dependent", out, w);
        run<2>("packed fma, independent x4", out, w);
        run<3>("packed fma, dependent", out, w);
    }
    return 0;
}
