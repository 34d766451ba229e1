// Does a SIMD of gfx950 issue scalar instructions of one wave beside vector instructions of another, or one instruction per turn?
// 1024 workgroups x 256 threads (4 waves per SIMD, every CU busy), each wave loops over 8 independent v_fma_f32 plus M scalar
// instructions (s_add_u32 on private SGPRs / taken branches / s_nop): if scalar issue rides along, time is flat in M until M ~ 8.
// Build: hipcc --offload-arch=gfx950 -O3 -o tools/issue_probe.out tools/experiments/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define V8 "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n" \
           "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
#define S1 "s_add_u32 %10, %10, 1\n"
#define S4 S1 S1 S1 S1
#define B1(n) "s_cbranch_scc0 1f\n s_nop 0\n 1:\n"
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
    float x[8];
    for (int j = 0; j < 8; j++) x[j] = threadIdx.x * 8 + j;
    uint32_t s = 0;
    for (int i = 0; i < iters; i++) {
#define OPS "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]), "+v"(x[4]), "+v"(x[5]), "+v"(x[6]), "+v"(x[7]) : "v"(a), "v"(b), "s"(s)
        if (MODE == 0) asm volatile(V8 : OPS);
        if (MODE == 1) asm volatile(V8 S4 : OPS : "scc");
        if (MODE == 2) asm volatile(V8 S4 S4 : OPS : "scc");
        if (MODE == 3) asm volatile(V8 S4 S4 S4 S4 : OPS : "scc");
        if (MODE == 4) asm volatile("v_fma_f32 %0, %0, %8, %9\n" S1 "v_fma_f32 %1, %1, %8, %9\n" S1 "v_fma_f32 %2, %2, %8, %9\n" S1 "v_fma_f32 %3, %3, %8, %9\n" S1
                                    "v_fma_f32 %4, %4, %8, %9\n" S1 "v_fma_f32 %5, %5, %8, %9\n" S1 "v_fma_f32 %6, %6, %8, %9\n" S1 "v_fma_f32 %7, %7, %8, %9\n" S1 : OPS : "scc");
        if (MODE == 5) asm volatile(V8 "s_cmp_eq_u32 %10, %10\n s_cbranch_scc1 1f\n s_nop 0\n 1:\n s_cbranch_scc1 2f\n s_nop 0\n 2:\n s_cbranch_scc1 3f\n s_nop 0\n 3:\n s_cbranch_scc1 4f\n s_nop 0\n 4:\n" : OPS : "scc");
        if (MODE == 6) asm volatile(V8 "s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n s_nop 0\n" : OPS);
        if (MODE == 7) asm volatile(V8 "v_readfirstlane_b32 %10, %0\n v_readfirstlane_b32 %10, %1\n v_readfirstlane_b32 %10, %2\n v_readfirstlane_b32 %10, %3\n" : OPS);
    }
    float r = 0;
    for (int j = 0; j < 8; j++) r += x[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r + float(s);
}
template <int MODE>
void run(const char* name, float* out, int wgs) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<wgs, 256>>>(out, 100, 1.0001f, 0.5f);
    hipEventRecord(e0);
    k<MODE><<<wgs, 256>>>(out, iters, 1.0001f, 0.5f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%4d workgroups  %-44s %.3f ms  = %.2f ns per loop body and SIMD-wave\n", wgs, name, ms, ms * 1e6 / iters / (wgs / 256.0));
}
int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    for (int wgs : {256, 512, 1024, 2048}) {
        run<0>("8 v_fma", out, wgs);
        run<1>("8 v_fma + 4 s_add (after)", out, wgs);
        run<2>("8 v_fma + 8 s_add (after)", out, wgs);
        run<3>("8 v_fma + 16 s_add (after)", out, wgs);
        run<4>("8 v_fma + 8 s_add (interleaved)", out, wgs);
        run<5>("8 v_fma + s_cmp + 4 taken branches", out, wgs);
        run<6>("8 v_fma + 8 s_nop", out, wgs);
        run<7>("8 v_fma + 4 v_readfirstlane", out, wgs);
    }
    return 0;
}
